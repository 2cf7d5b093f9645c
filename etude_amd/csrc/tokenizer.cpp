// TinyREMITokenizer glue on either side of the decoder, natively: notes + tempo map -> REMI events
// (etude/data/tokenizer.py:166-297), id sequence -> bars (:43-76), events -> notes with velocities (:300-496).
// The reference is per-song Python with O(notes x measures) scans and O(notes^2) searches; this restatement keeps its exact
// double arithmetic and tie-breaking (first minimum wins, stable sorts, dict insertion order, Python round(x, 4), numpy's
// pairwise mean) so that results are bit-identical, and uses a sorted-measure lookup where the scans are provably equivalent.
// Host code: nothing here touches the GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "../../include/etude_hip.h"
#include "common.h"

struct etd_tok {
  struct Measure { double bpm, start, end; int time_sig; };
  std::vector<Measure> ms;
  bool sorted_disjoint = false;    // starts ascending and every end <= next start: "first measure containing t" == binary search
};

namespace {
using Measure = etd_tok::Measure;
const int ALLOWED_DUR[10] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};                       // tokenizer.py:20
const double IDX_2_POS[8] = {0.0, 1.0 / 6, 1.0 / 4, 1.0 / 3, 1.0 / 2, 2.0 / 3, 3.0 / 4, 5.0 / 6};   // :19

// first measure with start <= t < end (the reference's linear scan, :233-235 / :387-391), or -1.  When the measures are sorted
// and disjoint at most one contains t, and it is the last one that starts at or before t.
int find_measure(const etd_tok& tk, double t) {
  const auto& ms = tk.ms;
  if (tk.sorted_disjoint) {
    size_t lo = 0, hi = ms.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (ms[mid].start <= t) lo = mid + 1; else hi = mid; }
    if (lo == 0) return -1;
    return t < ms[lo - 1].end ? (int)(lo - 1) : -1;
  }
  for (size_t i = 0; i < ms.size(); ++i)
    if (ms[i].start <= t && t < ms[i].end) return (int)i;
  return -1;
}

int map_duration(double duration_sec, double bpm) {          // :118-132
  if (duration_sec <= 0 || bpm <= 0) return ALLOWED_DUR[0];
  const double seconds_per_beat = 60.0 / bpm;
  const double per16 = seconds_per_beat / 4.0;
  const double d16 = duration_sec / per16;
  int best = ALLOWED_DUR[0];
  double bd = std::fabs((double)best - d16);
  for (int i = 1; i < 10; ++i) { const double d = std::fabs((double)ALLOWED_DUR[i] - d16); if (d < bd) { bd = d; best = ALLOWED_DUR[i]; } }
  return best;
}

// :135-152 with allow_triplet=False (the only way encode calls it): keys 0, 1/4, 1/2, 3/4, 1 -> 0, 2, 4, 6, 8
void compute_rel_pos(double onset, double ms, double me, int time_sig, int* pos_idx, bool* is_last) {
  static const double KEY[5] = {0.0, 0.25, 0.5, 0.75, 1.0};
  static const int IDX[5] = {0, 2, 4, 6, 8};
  double m_rel = (onset - ms) / (me - ms);
  m_rel = std::max(0.0, std::min(1.0, m_rel));
  const double inv = 1.0 / (double)time_sig;
  const int b_idx = (int)(m_rel / inv);
  double r = std::fmod(m_rel, inv);                           // Python float %: both operands >= 0 here
  const double b_rel_time = r / inv;
  int best = 0;
  double bd = std::fabs(KEY[0] - b_rel_time);
  for (int i = 1; i < 5; ++i) { const double d = std::fabs(KEY[i] - b_rel_time); if (d < bd) { bd = d; best = i; } }
  *pos_idx = b_idx * 8 + IDX[best];
  *is_last = *pos_idx >= 8 * time_sig;
}

struct NoteInfo { int pitch, duration; bool has_grace; int grace; };

double py_round4(double x) {                                   // Python round(x, 4): correctly rounded decimal, ties to even
  if (!std::isfinite(x)) return x;
  char buf[512];
  snprintf(buf, sizeof buf, "%.4f", x);
  return strtod(buf, nullptr);
}

double np_pairwise_sum(const double* a, long long n) {         // numpy's float64 add.reduce (np.mean's summation order)
  if (n < 8) {
    double res = 0.;
    for (long long i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    long long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long long n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

struct DNote {                     // a decoded note while it is being processed (:446-496)
  int pitch; double onset, offset; int velocity; bool is_grace; int main_pitch; int measure_idx;
};
}  // namespace

extern "C" int etd_tok_create(const etd_tempo_region* regions, int n_regions, etd_tok** out) {
  if (!out || n_regions < 0 || (n_regions > 0 && !regions)) ETD_FAIL(ETD_EINVAL, "tok_create: bad arguments");
  etd_tok* tk = new etd_tok();
  if (n_regions > 0) {                                        // _create_measures, tokenizer.py:166-229
    for (int ri = 0; ri < n_regions; ++ri) {
      const etd_tempo_region& rg = regions[ri];
      if (rg.n_downbeats <= 0) continue;
      const double seconds_per_beat = 60.0 / rg.bpm;
      const double bar_duration = (double)rg.time_sig * seconds_per_beat;
      for (int i = 0; i < rg.n_downbeats; ++i) {
        const double start = rg.downbeats[i];
        double end;
        if (i < rg.n_downbeats - 1) end = rg.downbeats[i + 1];
        else if (ri < n_regions - 1) end = regions[ri + 1].start;
        else end = start + bar_duration;
        tk->ms.push_back({rg.bpm, start, end, rg.time_sig});
      }
    }
    const etd_tempo_region& fr = regions[0];
    const etd_tempo_region& lr = regions[n_regions - 1];
    if (fr.n_downbeats <= 0 || lr.n_downbeats <= 0) { delete tk; ETD_FAIL(ETD_EINVAL, "tok_create: first / last tempo region has no downbeats (the reference raises IndexError)"); }
    const double fd = fr.downbeats[0], fdur = (60.0 / fr.bpm) * (double)fr.time_sig;
    tk->ms.insert(tk->ms.begin(), Measure{fr.bpm, fd - fdur, fd, fr.time_sig});
    const double ld = lr.downbeats[lr.n_downbeats - 1], ldur = (60.0 / lr.bpm) * (double)lr.time_sig;
    tk->ms.push_back({lr.bpm, ld + ldur, ld + 2 * ldur, lr.time_sig});
  }
  bool ok = true;
  for (size_t i = 0; i + 1 < tk->ms.size() && ok; ++i) ok = tk->ms[i].start <= tk->ms[i + 1].start && tk->ms[i].end <= tk->ms[i + 1].start && tk->ms[i].start <= tk->ms[i].end;
  if (!tk->ms.empty()) ok = ok && tk->ms.back().start <= tk->ms.back().end;
  tk->sorted_disjoint = ok;
  *out = tk;
  return ETD_OK;
}

extern "C" void etd_tok_destroy(etd_tok* tk) { delete tk; }

extern "C" int etd_tok_num_measures(const etd_tok* tk) { return tk ? (int)tk->ms.size() : 0; }

extern "C" int etd_tok_measures(const etd_tok* tk, double* start, double* end, double* bpm, int32_t* time_sig) {
  if (!tk) ETD_FAIL(ETD_EINVAL, "tok_measures: null");
  for (size_t i = 0; i < tk->ms.size(); ++i) {
    if (start) start[i] = tk->ms[i].start;
    if (end) end[i] = tk->ms[i].end;
    if (bpm) bpm[i] = tk->ms[i].bpm;
    if (time_sig) time_sig[i] = tk->ms[i].time_sig;
  }
  return ETD_OK;
}

// encode(): tokenizer.py:231-252 (_assign_notes), :78-116 (grace notes), :265-297
extern "C" int etd_tok_encode(const etd_tok* tk, const etd_note* notes_in, long long n, int with_grace_note, etd_event* out, long long cap,
                              long long* n_out) {
  if (!tk || n < 0 || (n > 0 && !notes_in) || !n_out) ETD_FAIL(ETD_EINVAL, "tok_encode: bad arguments");
  struct In { double onset, offset; int pitch; bool has_grace; int grace; };
  std::vector<In> notes((size_t)n);
  for (long long i = 0; i < n; ++i) notes[(size_t)i] = {notes_in[i].onset, notes_in[i].offset, notes_in[i].pitch, false, 0};
  if (with_grace_note && n > 0) {
    std::stable_sort(notes.begin(), notes.end(), [](const In& a, const In& b) { return a.onset < b.onset || (a.onset == b.onset && a.pitch < b.pitch); });
    std::vector<char> keep((size_t)n, 1);
    for (long long i = 0; i + 1 < n; ++i) {
      if (!keep[(size_t)i]) continue;
      for (long long j = i + 1; j < n; ++j) {
        const double diff = notes[(size_t)j].onset - notes[(size_t)i].onset;
        if (diff >= 0.1) break;
        const int pd = notes[(size_t)j].pitch - notes[(size_t)i].pitch;
        if (1e-6 < diff && diff < 0.1 && std::abs(pd) == 1) {
          notes[(size_t)j].has_grace = true;
          notes[(size_t)j].grace = notes[(size_t)i].pitch > notes[(size_t)j].pitch ? 1 : -1;
          keep[(size_t)i] = 0;
          break;
        }
      }
    }
    std::vector<In> kept;
    for (long long i = 0; i < n; ++i) if (keep[(size_t)i]) kept.push_back(notes[(size_t)i]);
    notes.swap(kept);
  }
  const size_t M = tk->ms.size();
  std::vector<std::map<int, std::vector<NoteInfo>>> chords(M);
  for (const In& nt : notes) {
    const int mi = find_measure(*tk, nt.onset);
    if (mi < 0) continue;
    const Measure& m = tk->ms[(size_t)mi];
    if (m.end == m.start || m.time_sig <= 0) ETD_FAIL(ETD_EINVAL, "tok_encode: degenerate measure %d (the reference divides by zero here)", mi);
    int pos; bool last;
    compute_rel_pos(nt.onset, m.start, m.end, m.time_sig, &pos, &last);
    const NoteInfo info{nt.pitch, map_duration(nt.offset - nt.onset, m.bpm), nt.has_grace, nt.grace};
    if (last && (size_t)mi + 1 < M) chords[(size_t)mi + 1][0].push_back(info);
    else if (!last) chords[(size_t)mi][pos].push_back(info);
  }
  long long k = 0;
  auto put = [&](int type, int value) { if (k < cap && out) out[k] = {type, value}; ++k; };
  for (size_t mi = 0; mi < M; ++mi) {
    put(ETD_EV_BAR, 1);
    for (auto& kv : chords[mi]) {
      std::vector<NoteInfo>& v = kv.second;
      std::stable_sort(v.begin(), v.end(), [](const NoteInfo& a, const NoteInfo& b) { return a.pitch > b.pitch; });
      std::vector<NoteInfo> uniq;
      for (const NoteInfo& x : v) {
        bool seen = false;
        for (const NoteInfo& y : uniq) if (y.pitch == x.pitch) { seen = true; break; }
        if (!seen) uniq.push_back(x);
      }
      put(ETD_EV_POS, kv.first);
      for (const NoteInfo& x : uniq) {
        if (x.has_grace) put(ETD_EV_GRACE, x.grace);
        put(ETD_EV_NOTE, x.pitch);
        put(ETD_EV_DURATION, x.duration);
      }
    }
    put(ETD_EV_BAR, 0);
  }
  *n_out = k;
  if (k > cap) ETD_FAIL(ETD_ENOMEM, "tok_encode: need room for %lld events", k);
  return ETD_OK;
}

// split_sequence_into_bars(): tokenizer.py:43-76.  bar b = out_ids[bar_offsets[b] .. bar_offsets[b+1])
extern "C" int etd_tok_split_bars(const int32_t* ids, long long n, int bar_bos_id, int bar_eos_id, int32_t* out_ids, long long cap_ids,
                                  long long* bar_offsets, long long cap_bars, long long* n_bars) {
  if (n < 0 || (n > 0 && !ids) || !n_bars) ETD_FAIL(ETD_EINVAL, "tok_split_bars: bad arguments");
  std::vector<std::vector<int32_t>> bars;
  if (bar_bos_id < 0 || bar_eos_id < 0) {
    if (n > 0) bars.emplace_back(ids, ids + n);              // (the reference returns the sequence unsplit, with a warning)
  } else {
    std::vector<int32_t> cur;
    bool in_bar = false;
    for (long long i = 0; i < n; ++i) {
      const int32_t t = ids[i];
      if (t == bar_bos_id) {
        if (in_bar && !cur.empty()) bars.push_back(cur);
        cur.assign(1, t);
        in_bar = true;
      } else if (t == bar_eos_id) {
        if (in_bar) { cur.push_back(t); bars.push_back(cur); cur.clear(); in_bar = false; }
      } else if (in_bar) {
        cur.push_back(t);
      }
    }
    if (in_bar && !cur.empty()) {
      if (cur.back() != bar_eos_id) cur.push_back(bar_eos_id);
      bars.push_back(cur);
    }
    std::vector<std::vector<int32_t>> good;
    for (auto& b : bars) if (b.size() > 1 && b.front() == bar_bos_id && b.back() == bar_eos_id) good.push_back(std::move(b));
    bars.swap(good);
  }
  long long tot = 0;
  for (auto& b : bars) tot += (long long)b.size();
  *n_bars = (long long)bars.size();
  if (tot > cap_ids || (long long)bars.size() + 1 > cap_bars || !out_ids || !bar_offsets)
    ETD_FAIL(ETD_ENOMEM, "tok_split_bars: need room for %lld ids in %lld bars", tot, (long long)bars.size());
  long long p = 0;
  for (size_t b = 0; b < bars.size(); ++b) {
    bar_offsets[b] = p;
    std::copy(bars[b].begin(), bars[b].end(), out_ids + p);
    p += (long long)bars[b].size();
  }
  bar_offsets[bars.size()] = p;
  return ETD_OK;
}

// decode_to_notes(): tokenizer.py:446-496 -> _process_glissandos :300-376 -> _assign_velocity :378-444 -> sort
extern "C" int etd_tok_decode(const etd_tok* tk, const etd_event* ev, long long n, const double* volume, long long n_volume, etd_note* out,
                              long long cap, long long* n_out) {
  if (!tk || n < 0 || (n > 0 && !ev) || !n_out || n_volume < 0) ETD_FAIL(ETD_EINVAL, "tok_decode: bad arguments");
  const auto& ms = tk->ms;
  const long long M = (long long)ms.size();
  std::vector<DNote> raw;
  {
    long long ei = 0, mi = 0;
    double cur_onset = 0.0;
    bool pend = false; int pend_v = 0;
    const Measure* cm = nullptr;
    while (ei < n) {
      const etd_event& e = ev[ei];
      if (e.type == ETD_EV_BAR && e.value == 1) { cm = mi < M ? &ms[(size_t)mi] : nullptr; ++mi; ++ei; continue; }
      if (!cm) { ++ei; continue; }
      const double mdur = mi < M ? ms[(size_t)mi].start - cm->start : 0.0;
      const double spb = mdur > 1e-6 ? mdur / (double)cm->time_sig : 60.0 / cm->bpm;
      if (e.type == ETD_EV_POS) {
        const int b_idx = (int)std::floor((double)e.value / 8.0);            // divmod(value, 8)
        const int b_rel = e.value - b_idx * 8;
        cur_onset = cm->start + (((double)b_idx + IDX_2_POS[b_rel]) * spb);
        ++ei; continue;
      }
      if (e.type == ETD_EV_GRACE) { pend = true; pend_v = e.value; ++ei; continue; }
      if (e.type == ETD_EV_NOTE) {
        if (ei + 1 < n && ev[ei + 1].type == ETD_EV_DURATION) {
          const double dur = (double)ev[ei + 1].value * (spb / 4.0);
          if (cm->start <= cur_onset && cur_onset < cm->end) raw.push_back({e.value, cur_onset, cur_onset + dur, 80, false, 0, -1});
          if (pend) {
            const double go = cur_onset - 0.05;
            if (cm->start <= go) raw.push_back({e.value + pend_v, go, cur_onset, 65, true, e.value, -1});
            pend = false;
          }
          ei += 2;
        } else {
          ++ei;
        }
        continue;
      }
      ++ei;
    }
  }
  // ---- _process_glissandos: runs of >= 3 grace notes within 1 s become a white- or black-key run.  The decoded grace notes
  // carry no 'grace_info', so the reference's direction test never fires and is_upward is always False (:318,:351).
  std::vector<DNote> notes;
  if (raw.size() < 3) {
    notes = raw;
  } else {
    std::vector<size_t> gidx;
    for (size_t i = 0; i < raw.size(); ++i) if (raw[i].is_grace) gidx.push_back(i);
    std::vector<char> removed(raw.size(), 0);
    std::vector<DNote> added;
    size_t i = 0;
    while (i < gidx.size()) {
      const size_t s0 = gidx[i];
      if (removed[s0]) { ++i; continue; }
      std::vector<size_t> win{s0};
      size_t k = i + 1;
      while (k < gidx.size()) {
        const double span = raw[gidx[k]].onset - raw[s0].onset;
        if (span > 1.0) break;
        win.push_back(gidx[k]);
        ++k;
      }
      if (win.size() >= 3) {
        for (size_t w : win) removed[w] = 1;
        std::set<double> mains;
        for (size_t w : win) mains.insert(raw[w].offset);
        for (size_t x = 0; x < raw.size(); ++x) if (!raw[x].is_grace && mains.count(raw[x].onset)) removed[x] = 1;
        const DNote& sn = raw[win.front()]; const DNote& en = raw[win.back()];
        const double t0 = sn.onset, t1 = en.offset;
        const int sp = sn.main_pitch, ep = en.main_pitch;
        auto white = [](int p) { int m = p % 12; if (m < 0) m += 12; return m == 0 || m == 2 || m == 4 || m == 5 || m == 7 || m == 9 || m == 11; };
        int wc = 0;
        for (size_t w : win) wc += white(raw[w].main_pitch) ? 1 : 0;
        const bool use_white = wc >= (int)win.size() - wc;
        const int lo = std::min(sp, ep), hi = std::max(sp, ep);
        std::vector<int> gl;
        for (int p = lo; p <= hi; ++p) if (white(p) == use_white) gl.push_back(p);
        std::reverse(gl.begin(), gl.end());
        if (gl.size() > 1) {
          const double nd = (t1 - t0) / (double)gl.size();
          for (size_t q = 0; q < gl.size(); ++q) {
            const double on = t0 + (double)q * nd;
            added.push_back({gl[q], on, on + 0.1, 80, false, 0, -1});
          }
        }
        i = k;
      } else {
        ++i;
      }
    }
    for (size_t x = 0; x < raw.size(); ++x) if (!removed[x]) notes.push_back(raw[x]);
    notes.insert(notes.end(), added.begin(), added.end());
  }
  // ---- _assign_velocity
  if (!notes.empty()) {
    std::vector<std::vector<size_t>> in_measure((size_t)M);
    for (size_t x = 0; x < notes.size(); ++x) {
      const int mi = find_measure(*tk, notes[x].onset);
      if (mi >= 0) { in_measure[(size_t)mi].push_back(x); notes[x].measure_idx = mi; }
    }
    for (long long mi = 0; mi < M; ++mi) {
      const auto& mn = in_measure[(size_t)mi];
      if (mn.empty()) continue;
      double base = 75;
      if (volume) {
        const long long s = (long long)(ms[(size_t)mi].start * 20.0), e = (long long)(ms[(size_t)mi].end * 20.0);   // int(): toward zero
        if (e > s && e <= n_volume) {
          long long a = s < 0 ? std::max(0LL, s + n_volume) : std::min(s, n_volume);      // numpy slice semantics for a negative start
          long long b = e < 0 ? std::max(0LL, e + n_volume) : std::min(e, n_volume);
          if (b > a) {
            const double avg = np_pairwise_sum(volume + a, b - a) / (double)(b - a);
            base = 60 + std::pow(avg, 0.5) * 40;
          } else {
            base = 75;
          }
        } else {
          base = 75;
        }
      } else {
        const size_t c = mn.size();
        base = c < 20 ? 70 : (c < 30 ? 80 : 90);
      }
      std::vector<double> keys; std::vector<std::vector<size_t>> groups;       // defaultdict in insertion order
      for (size_t x : mn) {
        const double key = py_round4(notes[x].onset);
        size_t g = 0;
        for (; g < keys.size(); ++g) if (keys[g] == key) break;
        if (g == keys.size()) { keys.push_back(key); groups.emplace_back(); }
        groups[g].push_back(x);
      }
      for (auto& g : groups) {
        std::stable_sort(g.begin(), g.end(), [&](size_t a, size_t b) { return notes[a].pitch > notes[b].pitch; });
        for (size_t j = 0; j < g.size(); ++j) {
          double vel = std::max(base - 10, base - (double)(j * 2));
          if (notes[g[j]].pitch > 90) vel -= 10;
          notes[g[j]].velocity = (int)std::max(0.0, std::min(127.0, vel));
        }
      }
    }
    for (size_t x = 0; x < notes.size(); ++x) {
      if (!notes[x].is_grace) continue;
      int gv = 65;
      for (size_t y = 0; y < notes.size(); ++y)
        if (std::fabs(notes[y].onset - notes[x].offset) < 1e-4 && notes[y].pitch == notes[x].main_pitch) { gv = notes[y].velocity - 15; break; }
      if (notes[x].pitch > 90) gv -= 10;
      notes[x].velocity = std::max(0, std::min(127, gv));
    }
  }
  std::stable_sort(notes.begin(), notes.end(), [](const DNote& a, const DNote& b) { return a.onset < b.onset || (a.onset == b.onset && a.pitch < b.pitch); });
  *n_out = (long long)notes.size();
  if ((long long)notes.size() > cap || (!out && !notes.empty())) ETD_FAIL(ETD_ENOMEM, "tok_decode: need room for %zu notes", notes.size());
  for (size_t x = 0; x < notes.size(); ++x) out[x] = {notes[x].onset, notes[x].offset, notes[x].pitch, notes[x].velocity};
  return ETD_OK;
}
