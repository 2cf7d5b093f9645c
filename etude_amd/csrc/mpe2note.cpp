// Frame-wise predictions -> notes, on the host, O(n_note * T).
// Replaces AMTAPC_Extractor._mpe2note (etude/data/extractor.py:256-418), whose per-frame Python
// neighbour scans take seconds per clip and would dominate once the model runs on the GPU.
//
// Numerics: the reference mixes Python floats (float64) with np.float32 array scalars; under
// numpy >= 2 (NEP 50) the three-point interpolation `i*hop -/+ hop*0.5*(a-b)/(c-b)` is evaluated in
// float32 (Python floats are "weak" and adopt the array scalar's dtype), all other times are float64.
// This file reproduces exactly that (pinned by tests/golden/mpe2note.json through the C ABI).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/etude_hip.h"
#include "common.h"

namespace {
struct Peak { long long loc; double time; };

// every frame of a plateau that is >= thr and strictly above the nearest different neighbour on
// both sides is a peak (extractor.py:267-296)
void find_peaks(const float* x, long long T, int stride, float thr, double hop_sec, std::vector<Peak>& out) {
  out.clear();
  long long s = 0;
  while (s < T) {
    long long e = s;
    const float v = x[s * stride];
    while (e + 1 < T && x[(e + 1) * stride] == v) ++e;
    const bool left = (s == 0) || (v > x[(s - 1) * stride]);
    const bool right = (e == T - 1) || (v > x[(e + 1) * stride]);
    if (left && right && v >= thr) {
      for (long long i = s; i <= e; ++i) {
        double t;
        if (i == 0 || i == T - 1) {
          t = (double)i * hop_sec;
        } else {
          const float a = x[(i - 1) * stride], b = x[(i + 1) * stride], c = x[i * stride];
          const float ih = (float)((double)i * hop_sec);
          const float hh = (float)(hop_sec * 0.5);
          if (a == b) t = (double)i * hop_sec;
          else if (a > b) t = (double)(ih - (hh * (a - b)) / (c - b));
          else t = (double)(ih + (hh * (b - a)) / (c - a));
        }
        out.push_back({i, t});
      }
    }
    s = e + 1;
  }
}
}  // namespace

extern "C" int etd_mpe2note_modes(const float* onset, const float* offset, const float* mpe, const int8_t* velocity, long long T,
                                  int n_note, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr,
                                  int note_min, int mode_velocity, int mode_offset, etd_note* out, long long cap, long long* n_out) {
  if (!onset || !offset || !mpe || !velocity || T < 0 || n_note <= 0 || !n_out) ETD_FAIL(ETD_EINVAL, "mpe2note: bad args");
  if (mode_velocity < 0 || mode_velocity > 1 || mode_offset < 0 || mode_offset > 2) ETD_FAIL(ETD_EINVAL, "mpe2note: unknown mode");
  const double hop_sec = (double)hop_sample / (double)sr;
  std::vector<etd_note> notes;
  std::vector<Peak> on, off;
  for (int j = 0; j < n_note; ++j) {
    find_peaks(onset + j, T, n_note, thred_onset, hop_sec, on);
    find_peaks(offset + j, T, n_note, thred_offset, hop_sec, off);
    size_t p = 0;          // first offset peak with loc > loc_on (onsets ascend, so p only moves forward)
    long long below = 0;   // scan cursor for mpe < thr
    for (size_t k = 0; k < on.size(); ++k) {
      const long long loc_on = on[k].loc;
      long long loc_next; double t_next;
      if (k + 1 < on.size()) { loc_next = on[k + 1].loc; t_next = on[k + 1].time; }
      else { loc_next = T; t_next = (double)(T - 1) * hop_sec; }
      while (p < off.size() && off[p].loc <= loc_on) ++p;
      const bool flag_off = p < off.size();
      long long loc_off = flag_off ? off[p].loc : loc_on + 1;
      double t_off = flag_off ? off[p].time : 0.0;
      if (loc_off > loc_next) { loc_off = loc_next; t_off = t_next; }
      if (below < loc_on + 1) below = loc_on + 1;
      while (below < loc_next && !(mpe[below * n_note + j] < thred_mpe)) ++below;
      const bool flag_mpe = below < loc_next;
      const long long loc_mpe = flag_mpe ? below : loc_on + 1;
      const double t_mpe = (double)loc_mpe * hop_sec;
      const int vel = (int)velocity[loc_on * n_note + j];
      double off_val;
      if (!flag_off && !flag_mpe) off_val = t_next;
      else if (flag_off && !flag_mpe) off_val = t_off;
      else if (!flag_off && flag_mpe) off_val = t_mpe;
      else if (mode_offset == ETD_M2N_OFFSET) off_val = t_off;                                  // extractor.py:391-393
      else if (mode_offset == ETD_M2N_LONGER) off_val = (loc_off >= loc_mpe) ? t_off : t_mpe;    // :394-399
      else off_val = (loc_off <= loc_mpe) ? t_off : t_mpe;                                       // "shorter" (the default), :400-404
      if (mode_velocity == ETD_M2N_VEL_ORG || vel > 0) notes.push_back({on[k].time, off_val, j + note_min, vel});   // "ignore_zero" drops velocity 0 (:405-409)
      const size_t n = notes.size();
      if (n > 1 && notes[n - 1].pitch == notes[n - 2].pitch && notes[n - 1].onset < notes[n - 2].offset)
        notes[n - 2].offset = notes[n - 1].onset;
    }
  }
  // sorted(sorted(by pitch), by onset): stable, ties keep pitch order (extractor.py:416)
  std::stable_sort(notes.begin(), notes.end(), [](const etd_note& a, const etd_note& b) { return a.pitch < b.pitch; });
  std::stable_sort(notes.begin(), notes.end(), [](const etd_note& a, const etd_note& b) { return a.onset < b.onset; });
  *n_out = (long long)notes.size();
  if ((long long)notes.size() > cap || (!out && !notes.empty())) ETD_FAIL(ETD_ENOMEM, "mpe2note: need room for %zu notes", notes.size());
  std::copy(notes.begin(), notes.end(), out);
  return ETD_OK;
}

// the reference's defaults: mode_velocity = "ignore_zero", mode_offset = "shorter" (extractor.py:256)
extern "C" int etd_mpe2note(const float* onset, const float* offset, const float* mpe, const int8_t* velocity, long long T,
                            int n_note, float thred_onset, float thred_offset, float thred_mpe, int hop_sample, int sr,
                            int note_min, etd_note* out, long long cap, long long* n_out) {
  return etd_mpe2note_modes(onset, offset, mpe, velocity, T, n_note, thred_onset, thred_offset, thred_mpe, hop_sample, sr, note_min,
                            ETD_M2N_VEL_IGNORE_ZERO, ETD_M2N_SHORTER, out, cap, n_out);
}
