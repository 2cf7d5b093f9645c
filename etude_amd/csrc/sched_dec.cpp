// Native scheduler of the Decode stage: the bar loop of EtudeDecoder.generate (etude/models/etude_decoder.py:246-354)
// for MANY independent (song, attribute tuple) jobs, run as concurrent device streams (continuous batching).
// Host logic only -- prompt assembly (:257-296), history window (:346-348), token budget (:301,:352); every forward
// pass goes through the C ABI of this library (etd_decoder_begin_bars / _step / _poll / _read_many).
// Keeping this loop out of Python removes ~50 us of interpreter work per (job, bar) from the serving path.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/etude_hip_debug.h"
#include "common.h"

namespace {
constexpr int SRC_CLASS_ID = 1, TGT_CLASS_ID = 2;   // etude/data/dataset.py:18-19

struct Pair { const int32_t* x; int xn; std::vector<int32_t> y; int a[4]; };

struct Job {
  const etd_job* j;
  int bar = 0; long long total = 0; int slot = -1; int limit = 0;
  int n_out_known = 0;                    // tokens of the current bar the host knows of (last poll; 1 right after the bar's prefill)
  std::vector<Pair> hist;                 // <= n_ctx most recent (X, Y, attrs)
  std::vector<int32_t> out;               // [n_bars_done, len_0.., tokens...] built at the end
  std::vector<std::vector<int32_t>> bars; // generated bars ([Bar_BOS] + tokens)
};

// prompt of the job's current bar appended to ids/cls/attr rows; returns its length
int assemble(const Job& jb, const etd_sched_cfg& c, std::vector<int32_t>& ids, std::vector<int32_t>& cls, std::vector<int32_t> (&at)[4]) {
  const size_t start = ids.size();
  auto seg = [&](const int32_t* p, int n, int cl, const int* a4) {
    ids.insert(ids.end(), p, p + n);
    cls.insert(cls.end(), (size_t)n, cl);
    for (int k = 0; k < 4; ++k) at[k].insert(at[k].end(), (size_t)n, a4[k]);
  };
  const int32_t empty[2] = {c.bar_bos_id, c.bar_eos_id};
  const int neutral[4] = {1, 1, 1, 1};                                   // etude_decoder.py:250
  const int nh = (int)jb.hist.size();
  for (int i = 0; i < c.n_ctx_pairs - nh; ++i) { seg(empty, 2, SRC_CLASS_ID, neutral); seg(empty, 2, TGT_CLASS_ID, neutral); }
  for (const Pair& p : jb.hist) { seg(p.x, p.xn, SRC_CLASS_ID, p.a); seg(p.y.data(), (int)p.y.size(), TGT_CLASS_ID, p.a); }
  const etd_job& j = *jb.j;
  const int32_t* xb = j.x_ids + j.x_offsets[jb.bar];
  const int xn = j.x_offsets[jb.bar + 1] - j.x_offsets[jb.bar];
  const int* ya = j.attrs4 + 4 * jb.bar;
  seg(xb, xn, SRC_CLASS_ID, ya);
  size_t len = ids.size() - start;
  if ((long long)len > (long long)c.max_position_embeddings - c.max_bar_token_limit) {   // etude_decoder.py:285-289
    const size_t keep = (size_t)((double)c.max_position_embeddings * c.context_overlap_ratio);
    if (len > keep) {
      const size_t drop = len - keep;
      ids.erase(ids.begin() + start, ids.begin() + start + drop);
      cls.erase(cls.begin() + start, cls.begin() + start + drop);
      for (int k = 0; k < 4; ++k) at[k].erase(at[k].begin() + start, at[k].begin() + start + drop);
      len = keep;
    }
  }
  ids.push_back(c.bar_bos_id); cls.push_back(TGT_CLASS_ID);
  for (int k = 0; k < 4; ++k) at[k].push_back(ya[k]);
  return (int)len + 1;
}
}  // namespace

// test hook (no GPU needed): the prompt the scheduler would build for one bar
extern "C" int etd_debug_assemble_prompt(const etd_sched_cfg* cfg, int n_hist, const int32_t* const* hx, const int32_t* hxn,
                                         const int32_t* const* hy, const int32_t* hyn, const int32_t* hattrs4, const int32_t* x, int xn,
                                         const int32_t* y_attrs4, int32_t* ids_out, int32_t* cls_out, int32_t* attrs4_out, int cap, int* T_out) {
  if (!cfg || n_hist < 0 || !x || !y_attrs4 || !ids_out || !cls_out || !attrs4_out || !T_out) ETD_FAIL(ETD_EINVAL, "assemble_prompt: bad arguments");
  if (cfg->struct_bytes != (int)sizeof(etd_sched_cfg)) ETD_FAIL(ETD_EINVAL, "assemble_prompt: etd_sched_cfg of %d bytes, expected %d", cfg->struct_bytes, (int)sizeof(etd_sched_cfg));
  const int32_t offs[2] = {0, xn};
  etd_job j{x, offs, 1, y_attrs4, nullptr};
  Job jb; jb.j = &j; jb.bar = 0;
  const int first = n_hist > cfg->n_ctx_pairs ? n_hist - cfg->n_ctx_pairs : 0;     // history_bar_pairs[-n:]
  for (int i = first; i < n_hist; ++i) {
    Pair p; p.x = hx[i]; p.xn = hxn[i]; p.y.assign(hy[i], hy[i] + hyn[i]); memcpy(p.a, hattrs4 + 4 * i, 16);
    jb.hist.push_back(std::move(p));
  }
  std::vector<int32_t> ids, cls, at[4];
  const int T = assemble(jb, *cfg, ids, cls, at);
  *T_out = T;
  if (T > cap) ETD_FAIL(ETD_ENOMEM, "assemble_prompt: need room for %d tokens", T);
  memcpy(ids_out, ids.data(), (size_t)T * 4); memcpy(cls_out, cls.data(), (size_t)T * 4);
  for (int k = 0; k < 4; ++k) memcpy(attrs4_out + (size_t)k * cap, at[k].data(), (size_t)T * 4);
  return ETD_OK;
}

extern "C" int etd_decoder_run_jobs(etd_dec* d, const etd_sched_cfg* cfg, const etd_job* jobs, int n_jobs, int32_t* out, long long out_cap,
                                    long long* job_offsets, long long* n_steps_out, void* stream) {
  if (!d || !cfg || !jobs || n_jobs < 1 || !out || !job_offsets) ETD_FAIL(ETD_EINVAL, "run_jobs: bad arguments");
  if (cfg->struct_bytes != (int)sizeof(etd_sched_cfg)) ETD_FAIL(ETD_EINVAL, "run_jobs: etd_sched_cfg of %d bytes, this library (ABI %d) expects %d -- caller built against another etude_hip.h", cfg->struct_bytes, ETD_ABI_VERSION, (int)sizeof(etd_sched_cfg));
  const etd_sched_cfg& c = *cfg;
  if (c.max_streams < 1 || c.n_ctx_pairs < 0 || c.max_bar_token_limit < 1 || c.max_prefill_rows < 1 || c.steps_per_poll < 1)
    ETD_FAIL(ETD_EINVAL, "run_jobs: bad scheduler config");
  if (!(c.temperature >= 0.f)) ETD_FAIL(ETD_EINVAL, "run_jobs: temperature must be >= 0");
  ETD_TRY(etd_decoder_set_sampling(d, c.temperature, c.top_p, c.seed, stream));
  std::vector<Job> J(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    J[i].j = &jobs[i];
    if (jobs[i].n_bars < 0 || (jobs[i].n_bars > 0 && (!jobs[i].x_ids || !jobs[i].x_offsets || !jobs[i].attrs4))) ETD_FAIL(ETD_EINVAL, "run_jobs: job %d malformed", i);
  }
  // a job whose upstream stages are still running is not touched before its flag is set (acquire: the flag's writer filled the bars first)
  auto is_ready = [&](int i) { return !jobs[i].ready || __atomic_load_n(jobs[i].ready, __ATOMIC_ACQUIRE) != 0; };
  std::vector<int> free_slots;
  for (int s = c.max_streams - 1; s >= 0; --s) free_slots.push_back(s);
  std::vector<int> active;          // job indices holding a slot
  int next_job = 0;
  long long n_steps = 0;
  std::vector<int32_t> ids, cls, at[4], a4cat, slots, Ts, tg, eos, lim, dn, no, rd, cnt;

  // ETD_SCHED_STATS=1: host time this engine spends with its queue EMPTY at bar boundaries (from the moment the poll reports
  // finished bars to the moment the next bars' launches have been issued), printed once per call
  const bool want_stats = ETD_XENV("ETD_SCHED_STATS") != nullptr;
  double host_gap_us = 0, poll_us = 0, read_us = 0, begin_us = 0; long long n_gaps = 0;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // start the next bar of the given jobs (prefill in as few passes as the row budget allows); jobs that are finished are retired
  auto start_bars = [&](std::vector<int>& batch) -> int {
    std::vector<int> pend;
    for (int ji : batch) {
      Job& jb = J[ji];
      bool fin = jb.bar >= jb.j->n_bars;
      if (!fin) {
        const long long cap = (long long)c.max_output_tokens - jb.total;
        jb.limit = (int)std::min<long long>(c.force_bar_tokens > 0 ? c.force_bar_tokens : c.max_bar_token_limit, cap);
        if (jb.limit <= 0) { jb.bars.push_back({c.bar_bos_id}); fin = true; }   // the reference's inner loop breaks before its first forward
      }
      if (fin) {
        free_slots.push_back(jb.slot); jb.slot = -1;
        active.erase(std::remove(active.begin(), active.end(), ji), active.end());
      } else {
        pend.push_back(ji);
      }
    }
    size_t p = 0;
    while (p < pend.size()) {
      ids.clear(); cls.clear(); for (auto& v : at) v.clear();
      slots.clear(); Ts.clear(); tg.clear(); eos.clear(); lim.clear();
      while (p < pend.size()) {
        Job& jb = J[pend[p]];
        const size_t before = ids.size();
        const int T = assemble(jb, c, ids, cls, at);
        if (!slots.empty() && (long long)ids.size() > c.max_prefill_rows) {   // does not fit this pass: undo, flush
          ids.resize(before); cls.resize(before); for (auto& v : at) v.resize(before);
          break;
        }
        slots.push_back(jb.slot); Ts.push_back(T);
        const int* ya = jb.j->attrs4 + 4 * jb.bar;
        tg.insert(tg.end(), ya, ya + 4);
        eos.push_back(c.force_bar_tokens > 0 ? -1 : c.bar_eos_id);
        lim.push_back(jb.limit);
        ++p;
      }
      if (c.temperature > 0.f) {           // draw key of a stream = (job, bar): the same tokens whatever slot / pass the job lands in
        std::vector<unsigned long long> keys(slots.size());
        size_t q = p - slots.size();
        const int kstride = c.job_key_stride > 0 ? c.job_key_stride : 1;
        for (size_t i = 0; i < slots.size(); ++i, ++q) keys[i] = ((unsigned long long)(uint32_t)(c.job_key_offset + pend[q] * kstride) << 32) | (uint32_t)J[pend[q]].bar;
        ETD_TRY(etd_decoder_set_keys(d, (int)slots.size(), slots.data(), keys.data()));
      }
      const size_t M = ids.size();
      a4cat.resize(4 * M);
      for (int k = 0; k < 4; ++k) memcpy(a4cat.data() + (size_t)k * M, at[k].data(), M * 4);
      const double tb0 = want_stats ? now_us() : 0;
      ETD_TRY(etd_decoder_begin_bars(d, (int)slots.size(), slots.data(), Ts.data(), ids.data(), cls.data(), a4cat.data(), tg.data(), eos.data(), lim.data(), stream));
      if (want_stats) begin_us += now_us() - tb0;
    }
    return ETD_OK;
  };

  bool skip_poll = false;           // the bars of every active stream were (re)started or polled since the last step launch: nothing new to learn from a poll
  while (next_job < n_jobs || !active.empty()) {
    std::vector<int> fresh;
    while (next_job < n_jobs && !free_slots.empty() && is_ready(next_job)) {
      Job& jb = J[next_job];
      jb.slot = free_slots.back(); free_slots.pop_back();
      active.push_back(next_job);
      fresh.push_back(next_job++);
    }
    if (!fresh.empty()) { ETD_TRY(start_bars(fresh)); for (int ji : fresh) J[ji].n_out_known = 1; }
    if (active.empty()) {
      if (next_job < n_jobs && !is_ready(next_job)) std::this_thread::sleep_for(std::chrono::microseconds(100));   // idle until upstream delivers
      continue;
    }
    const int na = (int)active.size();
    std::sort(active.begin(), active.end(), [&](int a, int b) { return J[a].slot < J[b].slot; });
    slots.resize(na); dn.resize(na); no.resize(na);
    for (int i = 0; i < na; ++i) slots[i] = J[active[i]].slot;
    const double t_poll0 = want_stats ? now_us() : 0;
    if (skip_poll) {
      // right after start_bars: a restarted stream has produced exactly its first token (the prefill's), the others are where
      // the last poll left them -- the step launches go out behind the prefill without a host round trip in between.  (A first
      // token that is already Bar_EOS just makes that row idle until the next poll.)
      for (int i = 0; i < na; ++i) { dn[i] = 0; no[i] = J[active[i]].n_out_known; }
      skip_poll = false;
    } else {
      ETD_TRY(etd_decoder_poll(d, slots.data(), na, dn.data(), no.data(), stream));
      for (int i = 0; i < na; ++i) J[active[i]].n_out_known = no[i];
    }
    const double t_poll1 = want_stats ? now_us() : 0;
    poll_us += t_poll1 - t_poll0;
    std::vector<int> done_jobs;
    for (int i = 0; i < na; ++i) if (dn[i]) done_jobs.push_back(active[i]);
    if (!done_jobs.empty()) {
      const int nd = (int)done_jobs.size(), cap = 1024;
      std::vector<int32_t> ds(nd);
      for (int i = 0; i < nd; ++i) ds[i] = J[done_jobs[i]].slot;
      rd.resize((size_t)nd * cap); cnt.resize(nd);
      ETD_TRY(etd_decoder_read_many(d, nd, ds.data(), rd.data(), cap, cnt.data(), stream));
      if (want_stats) read_us += now_us() - t_poll1;
      std::vector<int> again;
      for (int i = 0; i < nd; ++i) {
        Job& jb = J[done_jobs[i]];
        const int32_t* t = rd.data() + (size_t)i * cap;
        std::vector<int32_t> y; y.reserve(cnt[i] + 1);
        y.push_back(c.bar_bos_id); y.insert(y.end(), t, t + cnt[i]);
        jb.total += cnt[i];
        const etd_job& j = *jb.j;
        Pair pr; pr.x = j.x_ids + j.x_offsets[jb.bar]; pr.xn = j.x_offsets[jb.bar + 1] - j.x_offsets[jb.bar]; pr.y = y;
        memcpy(pr.a, j.attrs4 + 4 * jb.bar, 16);
        jb.hist.push_back(std::move(pr));
        if ((int)jb.hist.size() > c.n_ctx_pairs) jb.hist.erase(jb.hist.begin());
        jb.bars.push_back(std::move(y));
        jb.bar += 1;
        if (jb.total >= c.max_output_tokens) jb.bar = jb.j->n_bars;          // etude_decoder.py:352: stop after this bar
        again.push_back(done_jobs[i]);
      }
      ETD_TRY(start_bars(again));
      for (int ji : again) J[ji].n_out_known = 1;
      if (want_stats) { host_gap_us += now_us() - t_poll1; ++n_gaps; }
      skip_poll = true;
      continue;                                   // refill, then step without another poll
    }
    int nstep = c.steps_per_poll;
    if (c.force_bar_tokens > 0) {                 // no early EOS possible: run to the nearest bar end in one call
      nstep = 1 << 30;
      for (int i = 0; i < na; ++i) nstep = std::min(nstep, J[active[i]].limit - no[i]);
      nstep = std::max(nstep, 1);
    }
    ETD_TRY(etd_decoder_step(d, slots.data(), na, nstep, stream));
    n_steps += nstep;
  }
  if (want_stats) fprintf(stderr, "[sched] %lld bar boundaries: queue empty on the host side for %.0f us each on average (token read-back %.0f us, begin_bars call %.0f us, rest = prompt assembly); %.1f ms waiting in polls\n",
                          n_gaps, n_gaps ? host_gap_us / n_gaps : 0.0, n_gaps ? read_us / n_gaps : 0.0, n_gaps ? begin_us / n_gaps : 0.0, poll_us / 1e3);
  // pack results: per job [n_bars_done, len_0 .. len_{n-1}, tokens of bar 0, tokens of bar 1, ...]
  long long pos = 0;
  for (int i = 0; i < n_jobs; ++i) {
    job_offsets[i] = pos;
    const auto& bars = J[i].bars;
    long long need = 1 + (long long)bars.size();
    for (auto& b : bars) need += (long long)b.size();
    if (pos + need > out_cap) ETD_FAIL(ETD_ENOMEM, "run_jobs: output buffer too small (need > %lld ints)", out_cap);
    out[pos++] = (int32_t)bars.size();
    for (auto& b : bars) out[pos++] = (int32_t)b.size();
    for (auto& b : bars) { memcpy(out + pos, b.data(), b.size() * 4); pos += (long long)b.size(); }
  }
  job_offsets[n_jobs] = pos;
  if (n_steps_out) *n_steps_out = n_steps;
  return ETD_OK;
}
