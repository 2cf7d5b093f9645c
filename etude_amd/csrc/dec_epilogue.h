// Shared accumulator-tile epilogue of the decoder GEMM kernels (k_dgemm, k_dgemm_s and the big-tile k_linear
// decoder modes): bias / GELU / parallel-residual / logits / fused-QKV with RoPE + KV-cache scatter.
#pragma once
#include "dec_kernels.h"

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for the d16 pipeline, priced by ISSUE time: beside the MFMAs of a one-wave-per-SIMD kernel every VALU instruction costs its 4 clocks (a transcendental 8)
// and an MFMA gap hides only ~24 of them (LABNOTES round 4: k_dmlp_fused's chunks are issue-bound, sixteen GELUs per chunk).  x * Phi(x) with
// Phi(u) - 1/2 = u q(u^2), u = clamp(x, +-3.875), q a degree-6 minimax polynomial pinned to q(c^2) = 1 / (2 c) so that both tails are exact (0 and x):
// 10 plain VALU instructions = 40 clocks instead of 15 + two transcendentals = 76; |error| <= 6.8e-5 max(1, |x|) over all x (tools/fit_gelu.py), a
// quarter of the d16 rounding of the value it produces.  -DETD_GELU_AS keeps the round-2 form (erf by Abramowitz & Stegun 7.1.26, 1.5e-7) for A/B runs;
// the fp32 parity mode keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
#ifdef ETD_GELU_AS
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float er = fmaf(-p * t, e, 1.f);                  // erf(|x| / sqrt 2)
  return 0.5f * x * (1.f + copysignf(er, x));
#else
  const float u = __builtin_amdgcn_fmed3f(x, -3.875f, 3.875f);
  const float s = u * u;
  float q = fmaf(3.1433767589e-08f, s, -2.0315644633e-06f);
  q = fmaf(q, s, 5.6378066802e-05f); q = fmaf(q, s, -8.9401804144e-04f); q = fmaf(q, s, 9.1527355835e-03f);
  q = fmaf(q, s, -6.5392248333e-02f); q = fmaf(q, s, 3.9845609665e-01f);
  return x * fmaf(q, u, 0.5f);
#endif
}

// ---- shared epilogue: lane = token m, registers = features nb + acc_row(i, h)
template <bool WBF16, int EPI>
__device__ __forceinline__ void dgemm_epilogue(const DGemmArgs& a, const f32x16& acc, int m, int nb, int h) {
  // ---- epilogue: lane = token m, registers = features nb + acc_row(i, h)
  if constexpr (EPI == DEPI_QKV) {
    // fused QKV laid out [head][q|k|v][64] (modeling_gpt_neox.py:204-207)
    const int head = nb / 192, j0 = nb - head * 192, part = j0 >> 6, dbase = j0 & 63;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc[i] + a.bias[nb + acc_row(i, h)];
    const int pos = a.rows.pos[m];
    if (part < 2 && dbase == 0) {
      // partial RoPE on dims [0, 2*rot_half): pair (d, d + rot_half); with rot_half == 8 both sit in
      // this lane: d = (i&3) + 4h  (i < 4)  and d + 8 = register i + 4
      const float* cs = a.rope_cos + (long long)pos * a.rot_half;
      const float* sn = a.rope_sin + (long long)pos * a.rot_half;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = i + 4 * h;
        const float c = cs[d], s = sn[d];
        const float x1 = v[i], x2 = v[i + 4];
        v[i] = x1 * c - x2 * s;        // q*cos + rotate_half(q)*sin, first half:  x1*cos - x2*sin
        v[i + 4] = x2 * c + x1 * s;    // second half: x2*cos + x1*sin
      }
    }
    if (a.Qb && part == 0) {
      // batched prefill: d16 row-major Q for the MFMA attention kernel (K / V: the cache rows below)
      d16* dp = a.Qb + (long long)m * (a.n_heads * 64) + head * 64 + dbase;
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<d16x4*>(dp + 8 * q + 4 * h) = pack4d(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    if (part == 0) {
      if (!a.Qb) {
        float* qp = a.Q + (long long)m * (a.n_heads * 64) + head * 64 + dbase;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; *reinterpret_cast<f32x4*>(qp + 8 * q + 4 * h) = o; }
      }
    } else if (a.rows.active[m] && pos < a.max_ctx) {
      const long long off = (long long)a.rows.slot[m] * a.slot_stride + ((long long)head * a.max_ctx + pos) * 64 + dbase;
      void* base = part == 1 ? a.Kc : a.Vc;
      if constexpr (WBF16) {
        d16* kp = reinterpret_cast<d16*>(base) + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<d16x4*>(kp + 8 * q + 4 * h) = pack4d(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      } else {
        float* kp = reinterpret_cast<float*>(base) + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; *reinterpret_cast<f32x4*>(kp + 8 * q + 4 * h) = o; }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nb + 8 * q + 4 * h;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[4 * q + j];
        if (EPI != DEPI_LOGITS && EPI != DEPI_PARTIAL) v[j] += a.bias[n + j];
        if (EPI == DEPI_GELU) v[j] = WBF16 ? gelu_fast(v[j]) : gelu_erf(v[j]);
        if (EPI == DEPI_RELU) v[j] = fmaxf(v[j], 0.f);
      }
      if constexpr (EPI == DEPI_RESID) {
        const f32x4 ad = *reinterpret_cast<const f32x4*>(a.add + (long long)m * a.N + n);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(a.hin + (long long)m * a.N + n);
        const f32x4 o = {(v[0] + ad[0]) + hi[0], (v[1] + ad[1]) + hi[1], (v[2] + ad[2]) + hi[2], (v[3] + ad[3]) + hi[3]};
        *reinterpret_cast<f32x4*>(a.hout + (long long)m * a.N + n) = o;
      } else if (a.Yb) {     // d16 destination (feeds the next d16 GEMM)
        *reinterpret_cast<d16x4*>(a.Yb + (long long)m * a.ldy + n) = pack4d(v[0], v[1], v[2], v[3]);
      } else if (n + 3 < a.N) {
        const f32x4 o = {v[0], v[1], v[2], v[3]};
        float* yp = a.Y + (long long)m * a.ldy + n;
        if ((a.ldy & 3) == 0) *reinterpret_cast<f32x4*>(yp) = o;
        else { yp[0] = v[0]; yp[1] = v[1]; yp[2] = v[2]; yp[3] = v[3]; }
      } else {
        for (int j = 0; j < 4 && n + j < a.N; ++j) a.Y[(long long)m * a.ldy + n + j] = v[j];
      }
    }
  }
}

