// Frame-wise predictions -> notes ON THE DEVICE: the same arithmetic as mpe2note.cpp
// (AMTAPC_Extractor._mpe2note, etude/data/extractor.py:256-418), restructured so that nothing scans time sequentially:
//   * peak test per frame: a frame is a peak iff its plateau (run of equal values) is >= thr and strictly above the
//     nearest different neighbour on both sides (:267-296) -- each frame finds its own plateau ends;
//   * sub-frame time by the three-point interpolation in float32 (numpy >= 2 / NEP 50 semantics, see mpe2note.cpp);
//   * "first frame after the onset with mpe < thr" (:341-352): the reference's cursor never runs past the next onset,
//     so it restarts at loc_on + 1 for every onset => one backward "next frame below thr" scan per pitch answers all;
//   * offset peak after an onset (:328-340): binary search in the pitch's ordered offset-peak list;
//   * same-pitch overlap clipping (:411-414) only couples neighbours in emission order.
// One workgroup per pitch; ordered compaction keeps every list in time order, so the host only has to do the final
// (onset, pitch) stable sort of a few thousand notes.  Only the notes cross PCIe (~100 KB per clip instead of the
// 12.9 MB of frame-wise arrays).  Bit-exact with etd_mpe2note (tests/test_gpu_extractor.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/etude_hip.h"
#include "common.h"
#include "prof.h"

struct etd_m2n {
  int n_note = 0;
  long long Tcap = 0;
  int* pk_loc[2] = {nullptr, nullptr};       // [n_note][Tcap] onset / offset peak frames (ascending)
  double* pk_time[2] = {nullptr, nullptr};   // [n_note][Tcap]
  int* nb = nullptr;                         // [n_note][Tcap + 1] next frame >= t with mpe < thr (T if none)
  etd_note* notes = nullptr;                 // [n_note][Tcap] per-pitch notes in emission order
  int* cnt = nullptr;                        // [n_note] notes per pitch, [n_note] = total
  etd_note* packed = nullptr;                // [n_note * Tcap] notes of all pitches back to back (pitch-major)
  std::vector<etd_note> host;
};

namespace {
constexpr int TPB = 256;

struct M2nArgs {
  const float* onset; const float* offset; const float* mpe; const int8_t* vel;
  long long T; int n_note;
  float thr_on, thr_off, thr_mpe; double hop_sec; int note_min;
  int mode_velocity, mode_offset;            // ETD_M2N_VEL_* / ETD_M2N_* (etude_hip.h): extractor.py:256-258, 386-408
  int* pk_loc0; int* pk_loc1; double* pk_time0; double* pk_time1; int* nb; etd_note* notes; int* cnt; long long Tcap;
};

// is frame t of column x (stride n) a peak?  (extractor.py:267-296 / mpe2note.cpp find_peaks)
__device__ __forceinline__ bool is_peak(const float* x, long long T, int n, long long t, float thr, double hop_sec, double* time) {
  const float v = x[t * n];
  if (!(v >= thr)) return false;
  long long s = t, e = t;
  while (s > 0 && x[(s - 1) * n] == v) --s;
  while (e + 1 < T && x[(e + 1) * n] == v) ++e;
  const bool left = (s == 0) || (v > x[(s - 1) * n]);
  const bool right = (e == T - 1) || (v > x[(e + 1) * n]);
  if (!(left && right)) return false;
  double tt;
  if (t == 0 || t == T - 1) {
    tt = (double)t * hop_sec;
  } else {
    const float a = x[(t - 1) * n], b = x[(t + 1) * n], c = v;
    const float ih = (float)((double)t * hop_sec);
    const float hh = (float)(hop_sec * 0.5);
    if (a == b) tt = (double)t * hop_sec;
    else if (a > b) tt = (double)(ih - (hh * (a - b)) / (c - b));
    else tt = (double)(ih + (hh * (b - a)) / (c - a));
  }
  *time = tt;
  return true;
}

// block-wide ordered compaction step: returns this thread's slot (or -1) and advances *base (shared) by the number of flags
__device__ __forceinline__ int ordered_slot(bool flag, int* wsum /* [TPB/64] shared */, int* base /* shared */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(m);
  __syncthreads();
  int off = *base;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  int tot = 0;
  for (int w = 0; w < TPB / 64; ++w) tot += wsum[w];
  __syncthreads();
  if (threadIdx.x == 0) *base += tot;
  __syncthreads();
  return flag ? off + before : -1;
}

__global__ __launch_bounds__(TPB) void k_m2n_pitch(M2nArgs a) {
  __shared__ int wsum[TPB / 64];
  __shared__ int base;
  __shared__ int n_pk[2];
  __shared__ int carry[TPB];
  const int j = blockIdx.x, tid = threadIdx.x;
  const long long T = a.T;
  const int n = a.n_note;
  // ---- phases A, B: onset and offset peaks of this pitch, in time order
  for (int which = 0; which < 2; ++which) {
    const float* x = (which ? a.offset : a.onset) + j;
    const float thr = which ? a.thr_off : a.thr_on;
    int* loc = (which ? a.pk_loc1 : a.pk_loc0) + (long long)j * a.Tcap;
    double* tim = (which ? a.pk_time1 : a.pk_time0) + (long long)j * a.Tcap;
    if (tid == 0) base = 0;
    __syncthreads();
    for (long long t0 = 0; t0 < T; t0 += TPB) {
      const long long t = t0 + tid;
      double tt = 0.0;
      const bool pk = t < T && is_peak(x, T, n, t, thr, a.hop_sec, &tt);
      const int slot = ordered_slot(pk, wsum, &base);
      if (slot >= 0) { loc[slot] = (int)t; tim[slot] = tt; }
    }
    if (tid == 0) n_pk[which] = base;
    __syncthreads();
  }
  // ---- phase C: nb[t] = first frame >= t with mpe < thr (T if none); each thread owns a contiguous span, carries right to left
  int* nb = a.nb + (long long)j * (a.Tcap + 1);
  {
    const long long span = (T + TPB - 1) / TPB;
    const long long lo = (long long)tid * span, hi = (lo + span < T) ? lo + span : T;
    int first = (int)T;                                   // first below-threshold frame inside my span
    for (long long t = hi - 1; t >= lo; --t)
      if (a.mpe[t * n + j] < a.thr_mpe) first = (int)t;
    carry[tid] = first;
    __syncthreads();
    int after = (int)T;                                    // first below-threshold frame in any span to my right
    for (int w = tid + 1; w < TPB; ++w) { const int c = carry[w]; if (c < (int)T) { after = c; break; } }
    int cur = after;
    for (long long t = hi - 1; t >= lo; --t) {
      if (a.mpe[t * n + j] < a.thr_mpe) cur = (int)t;
      nb[t] = cur;
    }
    if (tid == 0) nb[T] = (int)T;
  }
  __threadfence_block();
  __syncthreads();
  // ---- phase D: one candidate note per onset peak (extractor.py:328-409), ordered compaction of those with velocity > 0
  const int n_on = n_pk[0], n_off = n_pk[1];
  const int* on_loc = a.pk_loc0 + (long long)j * a.Tcap; const double* on_t = a.pk_time0 + (long long)j * a.Tcap;
  const int* off_loc = a.pk_loc1 + (long long)j * a.Tcap; const double* off_t = a.pk_time1 + (long long)j * a.Tcap;
  etd_note* out = a.notes + (long long)j * a.Tcap;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n_on; k0 += TPB) {
    const int k = k0 + tid;
    bool emit = false;
    etd_note nt = {0.0, 0.0, 0, 0};
    if (k < n_on) {
      const long long loc_on = on_loc[k];
      long long loc_next; double t_next;
      if (k + 1 < n_on) { loc_next = on_loc[k + 1]; t_next = on_t[k + 1]; }
      else { loc_next = T; t_next = (double)(T - 1) * a.hop_sec; }
      int lo = 0, hi = n_off;                               // first offset peak with loc > loc_on
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (off_loc[mid] <= loc_on) lo = mid + 1; else hi = mid; }
      const bool flag_off = lo < n_off;
      long long loc_off = flag_off ? off_loc[lo] : loc_on + 1;
      double t_off = flag_off ? off_t[lo] : 0.0;
      if (loc_off > loc_next) { loc_off = loc_next; t_off = t_next; }
      const long long below = nb[loc_on + 1];                // loc_on + 1 <= T
      const bool flag_mpe = below < loc_next;
      const long long loc_mpe = flag_mpe ? below : loc_on + 1;
      const double t_mpe = (double)loc_mpe * a.hop_sec;
      const int vel = (int)a.vel[loc_on * n + j];
      double off_val;
      if (!flag_off && !flag_mpe) off_val = t_next;
      else if (flag_off && !flag_mpe) off_val = t_off;
      else if (!flag_off && flag_mpe) off_val = t_mpe;
      else if (a.mode_offset == ETD_M2N_OFFSET) off_val = t_off;                                   // (a) always the offset peak      extractor.py:387-389
      else if (a.mode_offset == ETD_M2N_LONGER) off_val = (loc_off >= loc_mpe) ? t_off : t_mpe;    // (b) the later of the two        :390-395
      else off_val = (loc_off <= loc_mpe) ? t_off : t_mpe;                                         // (c) "shorter", the default      :396-401
      emit = a.mode_velocity == ETD_M2N_VEL_ORG || vel > 0;                                        // "ignore_zero" drops velocity 0  :402-406
      nt.onset = on_t[k]; nt.offset = off_val; nt.pitch = j + a.note_min; nt.velocity = vel;
    }
    const int slot = ordered_slot(emit, wsum, &base);
    if (slot >= 0) out[slot] = nt;
  }
  __threadfence_block();
  __syncthreads();
  // ---- phase E: a note that starts before its same-pitch predecessor ends clips it (extractor.py:411-414)
  const int n_emit = base;
  for (int i = tid; i + 1 < n_emit; i += TPB) {
    const double nxt = out[i + 1].onset;
    if (nxt < out[i].offset) out[i].offset = nxt;
  }
  if (tid == 0) a.cnt[j] = n_emit;
}

// pitch-major concatenation of the per-pitch lists; cnt[n_note] = total
__global__ void k_m2n_pack(const etd_note* notes, int* cnt, int n_note, long long Tcap, etd_note* packed) {
  __shared__ int start[129];
  if (threadIdx.x == 0) {
    int s = 0;
    for (int j = 0; j < n_note; ++j) { start[j] = s; s += cnt[j]; }
    start[n_note] = s;
    cnt[n_note] = s;
  }
  __syncthreads();
  for (int j = blockIdx.x; j < n_note; j += gridDim.x) {
    const etd_note* src = notes + (long long)j * Tcap;
    const int c = start[j + 1] - start[j];
    for (int i = threadIdx.x; i < c; i += blockDim.x) packed[start[j] + i] = src[i];
  }
}
}  // namespace

extern "C" int etd_mpe2note_dev_create(int n_note, etd_m2n** out) {
  if (n_note < 1 || n_note > 128 || !out) ETD_FAIL(ETD_EINVAL, "mpe2note_dev_create: need 1 <= n_note <= 128");
  etd_m2n* h = new etd_m2n();
  h->n_note = n_note;
  *out = h;
  return ETD_OK;
}

extern "C" void etd_mpe2note_dev_destroy(etd_m2n* h) {
  if (!h) return;
  (void)hipDeviceSynchronize();
  for (int w = 0; w < 2; ++w) { (void)hipFree(h->pk_loc[w]); (void)hipFree(h->pk_time[w]); }
  (void)hipFree(h->nb); (void)hipFree(h->notes); (void)hipFree(h->cnt); (void)hipFree(h->packed);
  delete h;
}

static int m2n_reserve(etd_m2n* h, long long T) {
  if (T <= h->Tcap) return ETD_OK;
  (void)hipDeviceSynchronize();
  for (int w = 0; w < 2; ++w) { (void)hipFree(h->pk_loc[w]); (void)hipFree(h->pk_time[w]); h->pk_loc[w] = nullptr; h->pk_time[w] = nullptr; }
  (void)hipFree(h->nb); (void)hipFree(h->notes); (void)hipFree(h->cnt); (void)hipFree(h->packed);
  h->nb = nullptr; h->notes = nullptr; h->cnt = nullptr; h->packed = nullptr; h->Tcap = 0;
  const long long cap = ((T + 1023) / 1024) * 1024;
  const size_t n = (size_t)h->n_note * (size_t)cap;
  for (int w = 0; w < 2; ++w) { HIP_TRY(hipMalloc(&h->pk_loc[w], n * sizeof(int))); HIP_TRY(hipMalloc(&h->pk_time[w], n * sizeof(double))); }
  HIP_TRY(hipMalloc(&h->nb, (size_t)h->n_note * (size_t)(cap + 1) * sizeof(int)));
  HIP_TRY(hipMalloc(&h->notes, n * sizeof(etd_note)));
  HIP_TRY(hipMalloc(&h->packed, n * sizeof(etd_note)));
  HIP_TRY(hipMalloc(&h->cnt, (size_t)(h->n_note + 1) * sizeof(int)));
  h->Tcap = cap;
  return ETD_OK;
}

extern "C" int etd_mpe2note_dev(etd_m2n* h, const float* onset_dev, const float* offset_dev, const float* mpe_dev, const int8_t* vel_dev,
                                long long T, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr, int note_min,
                                etd_note* out, long long cap, long long* n_out, void* stream) {
  return etd_mpe2note_dev_modes(h, onset_dev, offset_dev, mpe_dev, vel_dev, T, thred_onset, thred_offset, thred_mpe, hop_sample, sr, note_min,
                                ETD_M2N_VEL_IGNORE_ZERO, ETD_M2N_SHORTER, out, cap, n_out, stream);
}

extern "C" int etd_mpe2note_dev_modes(etd_m2n* h, const float* onset_dev, const float* offset_dev, const float* mpe_dev, const int8_t* vel_dev,
                                      long long T, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr, int note_min,
                                      int mode_velocity, int mode_offset, etd_note* out, long long cap, long long* n_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!h || !onset_dev || !offset_dev || !mpe_dev || !vel_dev || T < 0 || T > 0x7ffffff0LL || !n_out || hop_sample < 1 || sr < 1 ||
      (mode_velocity != ETD_M2N_VEL_IGNORE_ZERO && mode_velocity != ETD_M2N_VEL_ORG) || mode_offset < ETD_M2N_SHORTER || mode_offset > ETD_M2N_OFFSET)
    ETD_FAIL(ETD_EINVAL, "mpe2note_dev: bad args");
  *n_out = 0;
  if (T == 0) return ETD_OK;
  ETD_TRY(m2n_reserve(h, T));
  M2nArgs a;
  a.onset = onset_dev; a.offset = offset_dev; a.mpe = mpe_dev; a.vel = vel_dev; a.T = T; a.n_note = h->n_note;
  a.thr_on = thred_onset; a.thr_off = thred_offset; a.thr_mpe = thred_mpe; a.hop_sec = (double)hop_sample / (double)sr; a.note_min = note_min;
  a.mode_velocity = mode_velocity; a.mode_offset = mode_offset;
  a.pk_loc0 = h->pk_loc[0]; a.pk_loc1 = h->pk_loc[1]; a.pk_time0 = h->pk_time[0]; a.pk_time1 = h->pk_time[1];
  a.nb = h->nb; a.notes = h->notes; a.cnt = h->cnt; a.Tcap = h->Tcap;
  {
    ProfScope ps("k_m2n_pitch", st, 0, (double)T * h->n_note * 13.0);
    hipLaunchKernelGGL(k_m2n_pitch, dim3(h->n_note), dim3(TPB), 0, st, a);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k_m2n_pack, dim3(h->n_note), dim3(256), 0, st, h->notes, h->cnt, h->n_note, h->Tcap, h->packed);
  HIP_TRY(hipGetLastError());
  int total = 0;
  HIP_TRY(hipMemcpyAsync(&total, h->cnt + h->n_note, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *n_out = total;
  if (total > cap || (!out && total > 0)) ETD_FAIL(ETD_ENOMEM, "mpe2note_dev: need room for %d notes", total);
  if (total == 0) return ETD_OK;
  h->host.resize((size_t)total);
  HIP_TRY(hipMemcpyAsync(h->host.data(), h->packed, (size_t)total * sizeof(etd_note), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // sorted(sorted(by pitch), by onset): stable, ties keep pitch order (extractor.py:416); the packed list is already pitch-major
  std::stable_sort(h->host.begin(), h->host.end(), [](const etd_note& x, const etd_note& y) { return x.onset < y.onset; });
  std::copy(h->host.begin(), h->host.end(), out);
  return ETD_OK;
}
