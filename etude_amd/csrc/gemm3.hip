// fp32-grade GEMM + attention on three f16 MFMAs per product tile -- see gemm3.h for the arithmetic and its measured error.
// Reference ops: F.linear / attention of etude/models/amt_apc.py:322-392 and HF modeling_gpt_neox.py:195-281, which the reference runs in fp32.
#include <cmath>
#include <cstring>
#include <vector>

#include "gemm3.h"
#include "dec_epilogue.h"
#include "prof.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// ================================================================================================ host: planes and bounds
int g3_pack_weights_host(const float* W, int N, int Npad, int K, uint16_t* dst) {
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)N * K; ++i) mx = fmaxf(mx, fabsf(W[i]));
  int ex = 0;
  if (mx > 0.f) (void)frexpf(mx, &ex);               // mx = f 2^ex, f in [0.5, 1)
  const int lg = mx > 0.f ? 15 - ex : 0;             // max |s w| = f 2^15 in [2^14, 2^15)
  const float s = ldexpf(1.f, lg);
  const int nchunk = K / 32;
  memset(dst, 0, g3_packed_elems(Npad, K) * 2);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float t = W[(size_t)n * K + k] * s;
      const f16 hi = (f16)t;
      const f16 lo = (f16)(t - (float)hi);
      const size_t blk = ((size_t)(n >> 7) * nchunk + (k >> 5)) * (2 * 128 * 32);
      const size_t in = (size_t)(n & 127) * 32 + (k & 31);
      memcpy(dst + blk + in, &hi, 2);
      memcpy(dst + blk + 128 * 32 + in, &lo, 2);
    }
  return lg;
}
float g3_bound_ln(const float* g, const float* b, int n) {
  float mx = 0.f;
  const float z = sqrtf((float)(n - 1));
  for (int k = 0; k < n; ++k) mx = fmaxf(mx, z * fabsf(g[k]) + fabsf(b[k]));
  return mx;
}
float g3_bound_linear_of_ln(const float* W, const float* c, int N, int K, const float* g, const float* b) {
  double mx = 0;
  for (int j = 0; j < N; ++j) {
    double q = 0, d = 0;
    for (int k = 0; k < K; ++k) { const double wg = (double)W[(size_t)j * K + k] * g[k]; q += wg * wg; d += (double)W[(size_t)j * K + k] * b[k]; }
    const double v = sqrt((double)K) * sqrt(q) + fabs(d) + (c ? fabs((double)c[j]) : 0.0);
    mx = v > mx ? v : mx;
  }
  return (float)mx;
}
void g3_row_bounds_of_ln(const float* W, const float* c, int N, int K, const float* g, const float* b, float* out) {
  for (int j = 0; j < N; ++j) out[j] = g3_bound_linear_of_ln(W + (size_t)j * K, c ? c + j : nullptr, 1, K, g, b);
}
float g3_bound_linear(const float* W, const float* c, int N, int K, float bx) {
  double mx = 0;
  for (int j = 0; j < N; ++j) {
    double l1 = 0;
    for (int k = 0; k < K; ++k) l1 += fabs((double)W[(size_t)j * K + k]);
    const double v = l1 * bx + (c ? fabs((double)c[j]) : 0.0);
    mx = v > mx ? v : mx;
  }
  return (float)mx;
}
int g3_scale_log2(float bound) {
  if (!(bound > 0.f) || !std::isfinite(bound)) return 0;
  int ex = 0;
  (void)frexpf(bound * 1.0001f, &ex);                // bound < 2^ex
  return 15 - ex;
}

// ================================================================================================ k_gemm3
// Workgroup = 4 waves = 128 tokens x 128 features, each wave 2 x 2 accumulator tiles of 32 x 32 (token on the lane: mfma(A = weight rows, B = token rows)), K in chunks
// of 32: the four planes of a chunk (X hi / lo converted from fp32 on the way in, W hi / lo copied from the packed stream) sit in 40 KiB of LDS as 80-byte rows (16
// consecutive rows start on 16 different 16-byte bank groups: conflict-free ds_read_b128 fragments); the next chunk's global loads are in flight behind the chunk's 24
// MFMAs; two or three workgroups per CU cover each other's barriers.
// KC = k per chunk (32; 64 is kept as a template instance for -DG3_KC_MAX=64 measurement builds).  LDS rows are KC + 8 elements (80 / 144 bytes: conflict-free ds_read_b128).
template <int KC> struct G3Geom { static constexpr int LDR = KC + 8, PLANE = 128 * LDR, NB = KC / 32; };

__device__ __forceinline__ f32x16 mfma16h(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// 8 fp32 values (already scaled) -> hi / lo f16 planes
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, float s, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t0 = v0[j] * s, t1 = v1[j] * s;
    const f16 h0 = (f16)t0, h1 = (f16)t1;
    hi[j] = h0; hi[4 + j] = h1;
    lo[j] = (f16)(t0 - (float)h0); lo[4 + j] = (f16)(t1 - (float)h1);
  }
}

// 4 fp32 values -> hi / lo f16
__device__ __forceinline__ void split4(const f32x4& v, float s, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float t = v[j] * s; const f16 x = (f16)t; hi[j] = x; lo[j] = (f16)(t - (float)x); }
}

// ---- the epilogue of one wave's 32-token x 64-feature block, rows first: the accumulator has the TOKEN on the lane, so a direct store writes 16 bytes into 32 different rows per
// instruction and the memory pipeline handles one row segment at a time (measured on the first version of this kernel: the K loop could be emptied of its MFMAs without
// the kernel getting faster).  The block goes through a wave-private LDS tile ([32][68] floats, the K-loop planes are free by then) and leaves as 256-byte row segments, four
// rows per instruction: bias / GELU / ReLU / RoPE on the registers before the transpose, residual reads and the Q / KV-cache scatter on the rows after it.
template <int EPI>
__device__ __forceinline__ void g3_store_block(const DGemmArgs& a, const f32x16 (&acc)[2], float inv, float* stg, int m_base, int p_M, int nb, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int m = m_base + r;
  int head = 0, part = 0;
  if constexpr (EPI == DEPI_QKV) { head = nb / 192; part = (nb - head * 192) >> 6; }
  int pos_l = 0;
  if constexpr (EPI == DEPI_QKV) pos_l = a.rows.pos[m < p_M ? m : p_M - 1];
#pragma unroll
  for (int tf = 0; tf < 2; ++tf) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = nb + tf * 32 + acc_row(i, h);
      v[i] = acc[tf][i] * inv;
      if (EPI != DEPI_LOGITS) v[i] += a.bias[n];          // (bias is padded to Npad)
      if (EPI == DEPI_GELU) v[i] = gelu_erf(v[i]);
      if (EPI == DEPI_RELU) v[i] = fmaxf(v[i], 0.f);
    }
    if constexpr (EPI == DEPI_QKV) {
      if (part < 2 && tf == 0) {
        // partial RoPE on dims [0, 16): pair (d, d + 8) = registers (i, i + 4), i < 4, d = i + 4 h        modeling_gpt_neox.py:111-151
        const float* cs = a.rope_cos + (long long)pos_l * a.rot_half;
        const float* sn = a.rope_sin + (long long)pos_l * a.rot_half;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float c = cs[i + 4 * h], sgn = sn[i + 4 * h];
          const float x1 = v[i], x2 = v[i + 4];
          v[i] = x1 * c - x2 * sgn;
          v[i + 4] = x2 * c + x1 * sgn;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      *reinterpret_cast<f32x4*>(stg + r * 68 + tf * 32 + 8 * q + 4 * h) = o;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int er = lane >> 4, ec = lane & 15;          // row-contiguous view: row = 4 it + er, 16-byte chunk ec (features 4 ec .. + 3 of the 64)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + er, mr = m_base + row;
    const f32x4 val = *reinterpret_cast<const f32x4*>(stg + row * 68 + ec * 4);
    if (mr >= p_M) continue;
    const int n = nb + ec * 4;
    if constexpr (EPI == DEPI_QKV) {
      if (part == 0) {
        *reinterpret_cast<f32x4*>(a.Q + (long long)mr * (a.n_heads * 64) + head * 64 + ec * 4) = val;
      } else {
        const int pos = a.rows.pos[mr];
        if (a.rows.active[mr] && pos < a.max_ctx) {
          float* base = reinterpret_cast<float*>(part == 1 ? a.Kc : a.Vc);
          *reinterpret_cast<f32x4*>(base + (long long)a.rows.slot[mr] * a.slot_stride + ((long long)head * a.max_ctx + pos) * 64 + ec * 4) = val;
        }
      }
    } else if constexpr (EPI == DEPI_RESID) {
      if (n < a.N) {
        const long long off = (long long)mr * a.N + n;
        f32x4 ad = {0.f, 0.f, 0.f, 0.f};
        if (a.add) ad = *reinterpret_cast<const f32x4*>(a.add + off);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(a.hin + off);
        const f32x4 o = {(val[0] + ad[0]) + hi[0], (val[1] + ad[1]) + hi[1], (val[2] + ad[2]) + hi[2], (val[3] + ad[3]) + hi[3]};
        *reinterpret_cast<f32x4*>(a.hout + off) = o;
      }
    } else {
      float* yp = a.Y + (long long)mr * a.ldy + n;
      if (n + 3 < a.N && (a.ldy & 3) == 0) *reinterpret_cast<f32x4*>(yp) = val;
      else { for (int j = 0; j < 4; ++j) if (n + j < a.N) yp[j] = val[j]; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                  // the tile is rewritten by the next block
}

// G3_ABL (measurement builds, ETD_FLAGS_GEMM3=-DG3_ABL=n; results are then wrong on purpose): 2 = no global loads inside the K loop, 4 = no MFMAs, 5 = no LDS stores
// inside the K loop, 6 = no epilogue stores
#ifndef G3_ABL
#define G3_ABL 0
#endif
template <int EPI, int KC>
__global__ __launch_bounds__(256, 2) void k_gemm3(const float* __restrict__ p_X, int p_ldx, const f16* __restrict__ p_Wp, int p_M, int p_K, float p_xs, float p_inv, DGemmArgs a) {
  using G = G3Geom<KC>;
  constexpr int LDR = G::LDR, PLANE = G::PLANE, NB = G::NB;
  __shared__ __attribute__((aligned(16))) f16 sm[4 * PLANE];        // X hi | X lo | W hi | W lo (>= 40 KiB: the epilogue's four [32][68] fp32 tiles fit)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  // workgroup -> tile, XCD-aware: consecutive workgroup ids go round the 8 XCDs, each with its own L2.  The NT column tiles of a row tile read the same 128 x K block of
  // X: they take CONSECUTIVE slots of ONE XCD (id = 8 * slot + xcd; slot = row-tile group * NT + column tile), so that the block comes from HBM once and from that L2
  // NT - 1 times -- with (row tile, column tile) = (blockIdx.x, blockIdx.y) the column tiles of a row tile ran 1 024 workgroups apart and X was read NT times from HBM
  const int NT = (int)gridDim.y, RT = (p_M + 127) >> 7;
  const int wid = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, slot = wid >> 3;
  const int nt = slot % NT, rt = (slot / NT) * 8 + (wid & 7);
  if (rt >= RT) return;
  const int m0 = rt * 128, nchunk = p_K / KC, nblk32 = p_K >> 5;
  // X staging: a load instruction covers FULL 128-byte row segments (8 lanes per row, 8 rows per wave, rows 32 apart per instruction) -- two lanes per row made every
  // instruction touch 32 different cache lines, and the CU's address path, not the bytes, was the limit
  // (row order: the four rows a 32-lane half of a store covers are 4 apart -- with 80-byte LDS rows they then start on bank offsets 0 / 64 / 128 / 192 of the 256-byte
  // bank cycle and the 8-byte hi / lo stores of the split are conflict-free; consecutive rows overlapped 48 bytes: a third of the LDS pipe's cycles were conflicts)
  const int xt = tid >> 3, xrow = (xt & 16) | ((xt & 3) << 2) | ((xt >> 2) & 3), xc = tid & 7;
  const float* xp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { int gm = m0 + xrow + 32 * i; gm = gm < p_M ? gm : p_M - 1; xp[i] = p_X + (long long)gm * p_ldx + xc * 4; }
  // W staging: the packed planes are contiguous per (tile, 32-block): 16 bytes per lane, 1 KiB per wave instruction
  // (row order: the eight rows a 16-lane pass of a 16-byte store covers are all even or all odd: their 32-byte pieces then fall on disjoint bank ranges)
  const int stt = tid >> 1, srow = (stt & ~15) | ((stt & 7) << 1) | ((stt >> 3) & 1), sh = tid & 1;
  const f16* wp = p_Wp + (size_t)nt * nblk32 * (2 * 128 * 32) + srow * 32 + sh * 16;
  f32x4 xr[NB][4]; u32x4 whr[NB][2], wlr[NB][2];
  auto gload = [&](int kc) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const f32x4*>(xp[i] + kc * KC + b * 32);
      const f16* w = wp + (size_t)(kc * NB + b) * (2 * 128 * 32);
      whr[b][0] = *reinterpret_cast<const u32x4*>(w); whr[b][1] = *reinterpret_cast<const u32x4*>(w + 8);
      wlr[b][0] = *reinterpret_cast<const u32x4*>(w + 128 * 32); wlr[b][1] = *reinterpret_cast<const u32x4*>(w + 128 * 32 + 8);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f16* Xh = sm; f16* Xl = sm + PLANE; f16* Wh = sm + 2 * PLANE; f16* Wl = sm + 3 * PLANE;
  const int so = srow * LDR + sh * 16, xo = xrow * LDR + xc * 4;
  const int fw = ((wave >> 1) * 64 + r) * LDR + h * 8, fx = ((wave & 1) * 64 + r) * LDR + h * 8;
  gload(0);
  for (int kc = 0; kc < nchunk; ++kc) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#if G3_ABL == 5
      if (kc > 0) continue;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f16x4 hi, lo;
        split4(xr[b][i], p_xs, hi, lo);
        *reinterpret_cast<f16x4*>(Xh + xo + 32 * i * LDR + b * 32) = hi; *reinterpret_cast<f16x4*>(Xl + xo + 32 * i * LDR + b * 32) = lo;
      }
      *reinterpret_cast<u32x4*>(Wh + so + b * 32) = whr[b][0]; *reinterpret_cast<u32x4*>(Wh + so + b * 32 + 8) = whr[b][1];
      *reinterpret_cast<u32x4*>(Wl + so + b * 32) = wlr[b][0]; *reinterpret_cast<u32x4*>(Wl + so + b * 32 + 8) = wlr[b][1];
    }
    __syncthreads();
#if G3_ABL != 2
    if (kc + 1 < nchunk) gload(kc + 1);
#endif
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      f16x8 wh[2], wl[2], xh[2], xl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        wh[t] = *reinterpret_cast<const f16x8*>(Wh + fw + t * 32 * LDR + ks * 16);
        wl[t] = *reinterpret_cast<const f16x8*>(Wl + fw + t * 32 * LDR + ks * 16);
        xh[t] = *reinterpret_cast<const f16x8*>(Xh + fx + t * 32 * LDR + ks * 16);
        xl[t] = *reinterpret_cast<const f16x8*>(Xl + fx + t * 32 * LDR + ks * 16);
      }
#pragma unroll
      for (int tf = 0; tf < 2; ++tf)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#if G3_ABL == 4
          acc[tf][tt][0] += (float)wl[tf][0] * (float)xh[tt][0] + (float)wh[tf][1] * (float)xl[tt][1];
#else
          acc[tf][tt] = mfma16h(wl[tf], xh[tt], acc[tf][tt]);
          acc[tf][tt] = mfma16h(wh[tf], xl[tt], acc[tf][tt]);
          acc[tf][tt] = mfma16h(wh[tf], xh[tt], acc[tf][tt]);
#endif
        }
    }
    __syncthreads();
  }
#if G3_ABL == 6
  if (acc[0][0][0] != 1.2345e30f) return;
#endif
  float* stg = reinterpret_cast<float*>(sm) + wave * (32 * 68);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const f32x16 blk[2] = {acc[0][tt], acc[1][tt]};
    g3_store_block<EPI>(a, blk, p_inv, stg, m0 + (wave & 1) * 64 + tt * 32, p_M, nt * 128 + (wave >> 1) * 64, lane);
  }
}

bool g3_enabled() { static const bool on = !ETD_XENV("ETD_NO_GEMM3"); return on; }

int launch_gemm3(const DGemmArgs& a, int epi, hipStream_t st) {
  if (a.M <= 0 || a.Npad % 128 || a.K % 32 || a.N > a.Npad || !a.Wp || !a.X || (a.ldx % 4) || ((uintptr_t)a.X & 15)) ETD_FAIL(ETD_EINVAL, "gemm3: bad shape M=%d N=%d Npad=%d K=%d", a.M, a.N, a.Npad, a.K);
  if (epi == DEPI_QKV && (a.rot_half != 8 || a.N % 192)) ETD_FAIL(ETD_EINVAL, "gemm3: QKV epilogue needs head_dim 64 and rotary_ndims 16");
  if (epi == DEPI_RESID && (a.N % 4)) ETD_FAIL(ETD_EINVAL, "gemm3: resid needs N %% 4 == 0");
  if (a.ln_g || a.Xb || a.Yb || a.Qb || a.k_splits > 1) ETD_FAIL(ETD_EINVAL, "gemm3: fp32 operands only (no fused LayerNorm, bf16 buffers or split-K)");
  ETD_LAUNCH_FILTER("k_gemm3");
  ProfScope ps(a.M >= 8192 ? "k_gemm3" : "k_gemm3_step", st, 2.0 * a.M * a.N * a.K, (double)a.Npad * a.K * 4 + (double)a.M * a.K * 4);
  const dim3 g((unsigned)((((a.M + 127) / 128 + 7) / 8) * 8), a.Npad / 128);      // row tiles rounded up to a multiple of 8 (one per XCD): see the kernel's workgroup -> tile map
  const float xs = ldexpf(1.f, a.x_log2), inv = ldexpf(1.f, -(a.x_log2 + a.w_log2));
// K per chunk: 32 (40 KiB of LDS, 154 registers: three workgroups per CU) measured 10 % faster than 64 (72 KiB, two per CU) on every shape of tools/bench_gemm3.py:
// what bounds this kernel is the dependent chain load -> split -> LDS -> barrier -> fragments -> MFMA of a workgroup, covered by the other workgroups of the CU, not a pipe
#ifndef G3_KC_MAX
#define G3_KC_MAX 32
#endif
#if G3_KC_MAX >= 64      // (the KC = 64 instances -- 72 KiB of static LDS, six epilogues -- exist only in such a measurement build)
  const bool k64 = a.K % 64 == 0;
#define G3_LAUNCH(E) do { if (k64) hipLaunchKernelGGL((k_gemm3<E, 64>), g, dim3(256), 0, st, a.X, a.ldx, (const f16*)a.Wp, a.M, a.K, xs, inv, a); \
                          else hipLaunchKernelGGL((k_gemm3<E, 32>), g, dim3(256), 0, st, a.X, a.ldx, (const f16*)a.Wp, a.M, a.K, xs, inv, a); } while (0)
#else
#define G3_LAUNCH(E) hipLaunchKernelGGL((k_gemm3<E, 32>), g, dim3(256), 0, st, a.X, a.ldx, (const f16*)a.Wp, a.M, a.K, xs, inv, a)
#endif
  switch (epi) {
    case DEPI_BIAS: G3_LAUNCH(DEPI_BIAS); break;
    case DEPI_GELU: G3_LAUNCH(DEPI_GELU); break;
    case DEPI_RELU: G3_LAUNCH(DEPI_RELU); break;
    case DEPI_RESID: G3_LAUNCH(DEPI_RESID); break;
    case DEPI_LOGITS: G3_LAUNCH(DEPI_LOGITS); break;
    case DEPI_QKV: G3_LAUNCH(DEPI_QKV); break;
    default: ETD_FAIL(ETD_EINVAL, "gemm3: unsupported epilogue %d", epi);
  }
#undef G3_LAUNCH
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================ k_gemm3_s
// The same arithmetic for a handful of rows (2 .. 512: decode steps of a small batch, the last-position tail of a prefill): k_dgemm_s's shape -- 32 tokens x 32 features
// per workgroup, K split over four waves, operand fragments straight from global memory / L2 to registers (the packed planes give a lane its 16 contiguous bytes per
// k-step), partial tiles reduced through LDS in a fixed order; optional LayerNorm over K fused in front (the row statistics are computed per workgroup, as k_dgemm_s does).
// What it replaces is that kernel's fp32 branch on v_mfma_f32_32x32x2_f32, whose K = 2048 chain alone took 31 us per launch at 54 rows.
#define G3S_WAVES 4
template <int EPI>
__global__ __launch_bounds__(64 * G3S_WAVES) void k_gemm3_s(int p_M, int p_Npad, int p_K, const f16* __restrict__ p_Wp, const float* __restrict__ p_X, int p_ldx, float p_xs, float p_inv, DGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float red[G3S_WAVES - 1][16][64];
  __shared__ float stat[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  // workgroup -> tile as in k_dgemm_s: the row tiles that stream the SAME weight tile take consecutive slots of ONE XCD (id = 8 * slot + xcd)
  const int RT = (p_M + 31) / 32, FT = p_Npad / 32;
  const int bid = blockIdx.x, wt = ((bid >> 3) / RT) * 8 + (bid & 7);
  if (wt >= FT) return;
  const int m0 = ((bid >> 3) % RT) * 32, n0 = wt * 32;
  const bool ln = a.ln_g != nullptr;
  if (ln) {
    for (int rr = 0; rr < 32 / G3S_WAVES; ++rr) {
      const int row = wave * (32 / G3S_WAVES) + rr;
      int gm = m0 + row; gm = gm < p_M ? gm : p_M - 1;
      const float* xp = p_X + (long long)gm * p_ldx;
      float s = 0.f;
      for (int k = lane * 4; k < p_K; k += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k); s = (((s + v[0]) + v[1]) + v[2]) + v[3]; }
      s = wave_sum(s);
      const float mean = s / (float)p_K;
      float q = 0.f;
      for (int k = lane * 4; k < p_K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dl = v[e] - mean; q = fmaf(dl, dl, q); }
      }
      q = wave_sum(q);
      if (lane == 0) { stat[row] = mean; stat[32 + row] = 1.f / sqrtf(q / (float)p_K + a.ln_eps); }
    }
    __syncthreads();
  }
  int gm = m0 + r; gm = gm < p_M ? gm : p_M - 1;
  const float mean = ln ? stat[r] : 0.f, rstd = ln ? stat[32 + r] : 1.f;
  const float* xrow = p_X + (long long)gm * p_ldx + h * 8;
  const int nchunk = p_K >> 5, n = n0 + r;
  const f16* wrow = p_Wp + (size_t)(n >> 7) * nchunk * (2 * 128 * 32) + (n & 127) * 32 + h * 8;      // + chunk * 8192 + (k & 31) [+ 4096: lo plane]
  const int kq = p_K / G3S_WAVES, kb = wave * kq, ke = kb + kq;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = kb; k < ke; k += 128) {                    // 8 k-steps per round trip: 16 weight fragments + 16 x 16 B of activations in flight
    f16x8 wh[8], wl[8]; f32x4 x0[8], x1[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int kk = k + s8 * 16;
      const f16* w = wrow + (size_t)(kk >> 5) * (2 * 128 * 32) + (kk & 31);
      wh[s8] = *reinterpret_cast<const f16x8*>(w);
      wl[s8] = *reinterpret_cast<const f16x8*>(w + 128 * 32);
      x0[s8] = *reinterpret_cast<const f32x4*>(xrow + kk);
      x1[s8] = *reinterpret_cast<const f32x4*>(xrow + kk + 4);
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      if (ln) {
        const int kk = k + s8 * 16 + h * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + kk), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + kk + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + kk), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + kk + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[s8][j] = (x0[s8][j] - mean) * rstd * g0[j] + b0[j]; x1[s8][j] = (x1[s8][j] - mean) * rstd * g1[j] + b1[j]; }
      }
      f16x8 xh, xl;
      split8(x0[s8], x1[s8], p_xs, xh, xl);
      acc = mfma16h(wl[s8], xh, acc);
      acc = mfma16h(wh[s8], xl, acc);
      acc = mfma16h(wh[s8], xh, acc);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < G3S_WAVES - 1; ++w)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
  const int m = m0 + r;
  if (m >= p_M) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] *= p_inv;
  dgemm_epilogue<false, EPI>(a, acc, m, n0, h);
}

// Which kernel: the 32 x 32-tile kernel up to 512 rows, and up to 2 048 rows when the output is at most 512 wide -- there k_gemm3's 128 x 128 tiles are a handful of
// workgroups walking the whole K one after the other (measured, tools/bench_gemm3.py step: attention.dense 21.8 -> 12.1 us at 576 rows, 22.3 -> 20.3 at 1 728; the
// K = 2 048 down projection 70.9 -> 29.5 and 71.7 -> 54.7; QKV and up, 1 536 / 2 048 wide, are faster on the tiles from 576 rows up).  ETD_G3S_MAX_ROWS overrides the row limit.
static int g3s_max_rows(int npad) {
  static const int v = ETD_XENV("ETD_G3S_MAX_ROWS") ? atoi(ETD_XENV("ETD_G3S_MAX_ROWS")) : 0;
  return v > 0 ? v : (npad <= 512 ? 2048 : 512);
}
bool gemm3_s_takes(const DGemmArgs& a, int epi) {
  return a.Wp && a.M >= 2 && a.M <= g3s_max_rows(a.Npad) && a.K % (128 * G3S_WAVES) == 0 && a.Npad % 128 == 0 && epi != DEPI_PARTIAL && !a.Xb && !a.Yb && !a.Qb && a.k_splits <= 1;
}
int launch_gemm3_s(const DGemmArgs& a, int epi, hipStream_t st) {
  if (!gemm3_s_takes(a, epi) || !a.X || (a.ldx % 4) || ((uintptr_t)a.X & 15) || a.N > a.Npad) ETD_FAIL(ETD_EINVAL, "gemm3_s: bad shape M=%d N=%d Npad=%d K=%d", a.M, a.N, a.Npad, a.K);
  if (epi == DEPI_QKV && (a.rot_half != 8 || a.N % 192)) ETD_FAIL(ETD_EINVAL, "gemm3_s: QKV epilogue needs head_dim 64 and rotary_ndims 16");
  if (epi == DEPI_RESID && (a.N % 4)) ETD_FAIL(ETD_EINVAL, "gemm3_s: resid needs N %% 4 == 0");
  ETD_LAUNCH_FILTER("k_gemm3_s");
  ProfScope ps("k_gemm3_s", st, 2.0 * a.M * a.N * a.K, (double)a.Npad * a.K * 4);
  const int wtiles = a.Npad / 32;
  const dim3 g((unsigned)(((wtiles + 7) / 8) * 8 * ((a.M + 31) / 32)));
  const float xs = ldexpf(1.f, a.x_log2), inv = ldexpf(1.f, -(a.x_log2 + a.w_log2));
#define G3S_LAUNCH(E) hipLaunchKernelGGL((k_gemm3_s<E>), g, dim3(64 * G3S_WAVES), 0, st, a.M, a.Npad, a.K, (const f16*)a.Wp, a.X, a.ldx, xs, inv, a)
  switch (epi) {
    case DEPI_BIAS: G3S_LAUNCH(DEPI_BIAS); break;
    case DEPI_GELU: G3S_LAUNCH(DEPI_GELU); break;
    case DEPI_RELU: G3S_LAUNCH(DEPI_RELU); break;
    case DEPI_RESID: G3S_LAUNCH(DEPI_RESID); break;
    case DEPI_LOGITS: G3S_LAUNCH(DEPI_LOGITS); break;
    case DEPI_QKV: G3S_LAUNCH(DEPI_QKV); break;
    default: ETD_FAIL(ETD_EINVAL, "gemm3_s: unsupported epilogue %d", epi);
  }
#undef G3S_LAUNCH
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================ fp32 LayerNorm rows
// one wave per row; mean and biased variance over H as F.layer_norm computes them, (x - mean) / sqrt(var + eps) * g + b
template <int NV>      // H = 256 NV
__global__ __launch_bounds__(256) void k_ln_rows_f32(const float* __restrict__ hsrc, int M, const float* __restrict__ g1, const float* __restrict__ b1,
                                                     const float* __restrict__ g2, const float* __restrict__ b2, float eps, float* __restrict__ x1, float* __restrict__ x2) {
  constexpr int H = 256 * NV;
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) { v[j] = *reinterpret_cast<const f32x4*>(hsrc + (long long)m * H + j * 256 + lane * 4); s = (((s + v[j][0]) + v[j][1]) + v[j][2]) + v[j][3]; }      // (a serial chain: the SLP vectoriser must not pair these crosswise -- tests/test_isa_guard.py)
  s = wave_sum(s);
  const float mean = s * (1.f / H);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float dl = v[j][e] - mean; q = fmaf(dl, dl, q); }
  q = wave_sum(q);
  const float rstd = 1.f / sqrtf(q * (1.f / H) + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = j * 256 + lane * 4;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(g1 + k), ba = *reinterpret_cast<const f32x4*>(b1 + k);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * ga[e] + ba[e];
    *reinterpret_cast<f32x4*>(x1 + (long long)m * H + k) = o;
    if (x2) {
      const f32x4 gb = *reinterpret_cast<const f32x4*>(g2 + k), bb = *reinterpret_cast<const f32x4*>(b2 + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gb[e] + bb[e];
      *reinterpret_cast<f32x4*>(x2 + (long long)m * H + k) = o;
    }
  }
}
int launch_ln_rows_f32(const float* h, int M, int H, const float* g1, const float* b1, const float* g2, const float* b2, float eps, float* x1, float* x2, hipStream_t st) {
  if (M <= 0 || H % 256 || H > 1024 || !h || !g1 || !b1 || !x1 || (x2 && (!g2 || !b2))) ETD_FAIL(ETD_EINVAL, "ln_rows_f32: bad arguments");
  ProfScope ps("k_ln_rows_f32", st, 0.0, (double)M * H * 4 * (x2 ? 3 : 2));
  const dim3 g((M + 3) / 4);
  switch (H / 256) {
    case 1: hipLaunchKernelGGL(k_ln_rows_f32<1>, g, dim3(256), 0, st, h, M, g1, b1, g2, b2, eps, x1, x2); break;
    case 2: hipLaunchKernelGGL(k_ln_rows_f32<2>, g, dim3(256), 0, st, h, M, g1, b1, g2, b2, eps, x1, x2); break;
    case 3: hipLaunchKernelGGL(k_ln_rows_f32<3>, g, dim3(256), 0, st, h, M, g1, b1, g2, b2, eps, x1, x2); break;
    default: hipLaunchKernelGGL(k_ln_rows_f32<4>, g, dim3(256), 0, st, h, M, g1, b1, g2, b2, eps, x1, x2); break;
  }
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================ k_attn3
// S^T = K Q^T with the QUERY on the lane (online-softmax state per lane, one exchange with lane ^ 32), P^T feeds the PV MFMA from the accumulator registers, V^T by
// ds_read_b64_tr_b16 from a key-major LDS tile -- the skeleton of k_pattn (csrc/dec_prefill.hip) -- with every operand as hi / lo f16 planes and three MFMAs per
// product: Q planes in registers (converted once), K / V tiles converted from fp32 on their way into LDS, P split from the fp32 accumulator.  P in [0, 1] is carried
// at 2^15.  Workgroup = NW waves x 32 queries of one (sequence, head), key tiles of 64.
#define A3_LDK 72     // K plane row stride (elements): 144 B
#define A3_LDV 96     // V plane row stride (elements): 192 B (k_pattn: the four key rows of a transposing read on four bank groups)
typedef __attribute__((ext_vector_type(4))) short a3_s16x4;
typedef __attribute__((address_space(3))) a3_s16x4* a3_lds_s16x4;

template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void k_attn3(Attn3Args a) {
  __shared__ __attribute__((aligned(16))) f16 sm[2 * 64 * A3_LDK + 2 * 64 * A3_LDV];
  f16* Kh = sm; f16* Kl = Kh + 64 * A3_LDK; f16* Vh = Kl + 64 * A3_LDK; f16* Vl = Vh + 64 * A3_LDV;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int nh = a.n_heads, qpw = (int)(blockDim.x >> 6) * 32;        // queries per workgroup
  const int seq = blockIdx.y / nh, head = blockIdx.y - seq * nh;
  int Sq, Sk, off = 0;
  const float *qb, *kb, *vb; float* ob;
  long long ldq = a.ldq, ldk = a.ldk, ldv = a.ldv, ldo = a.ldo;
  if constexpr (RAGGED) {
    Sq = Sk = a.seq_len[seq];
    const long long r0 = a.seq_row0[seq];
    off = (qpw - (Sq % qpw)) % qpw;                                    // query tiles aligned to the END of the prompt (k_pattn)
    if ((int)blockIdx.x * qpw - off >= Sq) return;
    const int slot = a.row_slot[r0];
    const long long cbase = (long long)slot * a.slot_stride + (long long)head * a.max_ctx * 64;
    qb = a.Q + r0 * ldq + head * 64; ob = a.O + r0 * ldo + head * 64;
    kb = a.K + cbase; vb = a.V + cbase; ldk = ldv = 64;
  } else {
    Sq = a.Sq; Sk = a.Sk;
    if ((int)blockIdx.x * qpw >= Sq) return;
    qb = a.Q + seq * a.q_seq + head * 64; ob = a.O + seq * a.o_seq + head * 64;
    kb = a.K + seq * a.k_seq + head * 64; vb = a.V + seq * a.v_seq + head * 64;
  }
  const int q0 = (int)blockIdx.x * qpw + wave * 32 - off;            // wave-uniform
  int qi = q0 + r; const bool qvalid = qi >= 0 && qi < Sq; qi = qi < 0 ? 0 : (qi < Sq ? qi : Sq - 1);

  const float qs = ldexpf(1.f, a.q_log2), ksc = ldexpf(1.f, a.k_log2), vsc = ldexpf(1.f, a.v_log2);
  f16x8 qh[4], ql[4];
  {
    const float* qp = qb + (long long)qi * ldq + h * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + s * 16), v1 = *reinterpret_cast<const f32x4*>(qp + s * 16 + 4);
      split8(v0, v1, qs, qh[s], ql[s]);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  int ntile = (Sk + 63) >> 6;
  if constexpr (RAGGED) { const int lim = ((int)blockIdx.x * qpw + qpw - 1 - off) / 64 + 1; ntile = ntile < lim ? ntile : lim; }
  // K / V tiles: 64 keys x 64 d fp32 each = 1024 16-byte chunks per tensor, 4 per thread at 256 threads (16 chunks per row)
  f32x4 kreg[6], vreg[6];                  // up to 6 chunks per thread (192-thread launches: 1024 / 192 -> 6 rounds, the last partly idle)
  const int nround = (1024 + nthr - 1) / nthr;
  auto tile_gload = [&](int kv0_) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < nround) {
        const int c = tid + i * nthr;
        if (c < 1024) {
          const int row = c >> 4, ch = c & 15;
          int key = kv0_ + row; key = key < Sk ? key : Sk - 1;       // rows past the end: a finite copy of the last key (their P is 0)
          kreg[i] = *reinterpret_cast<const f32x4*>(kb + (long long)key * ldk + ch * 4);
          vreg[i] = *reinterpret_cast<const f32x4*>(vb + (long long)key * ldv + ch * 4);
        }
      }
    }
  };
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
  const int vtr = (4 * h + tq) * A3_LDV + 16 * tg + 4 * tp;
  // exponent base: scores in accumulator units (s_true 2^(q_log2 + k_log2)); c folds the plane scales, 1 / sqrt(d) and log2(e)
  const float c = a.scale * 1.4426950408889634f * ldexpf(1.f, -(a.q_log2 + a.k_log2));
  tile_gload(0);
  for (int jt = 0; jt < ntile; ++jt) {
    const int kv0 = jt * 64;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < nround) {
        const int cidx = tid + i * nthr;
        if (cidx < 1024) {
          const int row = cidx >> 4, ch = cidx & 15;
          f16x4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float t = kreg[i][j] * ksc; const f16 x = (f16)t; hi[j] = x; lo[j] = (f16)(t - (float)x); }
          *reinterpret_cast<f16x4*>(Kh + row * A3_LDK + ch * 4) = hi; *reinterpret_cast<f16x4*>(Kl + row * A3_LDK + ch * 4) = lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float t = vreg[i][j] * vsc; const f16 x = (f16)t; hi[j] = x; lo[j] = (f16)(t - (float)x); }
          *reinterpret_cast<f16x4*>(Vh + row * A3_LDV + ch * 4) = hi; *reinterpret_cast<f16x4*>(Vl + row * A3_LDV + ch * 4) = lo;
        }
      }
    }
    __syncthreads();
    if (jt + 1 < ntile) tile_gload(kv0 + 64);
    if (!RAGGED || kv0 <= q0 + 31) {          // (wave-uniform) causal: some query of this wave sees a key of this tile
      f32x16 sT[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sT[kt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f16x8 kh = *reinterpret_cast<const f16x8*>(Kh + (kt * 32 + r) * A3_LDK + s * 16 + h * 8);
          const f16x8 kl = *reinterpret_cast<const f16x8*>(Kl + (kt * 32 + r) * A3_LDK + s * 16 + h * 8);
          sT[kt] = mfma16h(kl, qh[s], sT[kt]);
          sT[kt] = mfma16h(kh, ql[s], sT[kt]);
          sT[kt] = mfma16h(kh, qh[s], sT[kt]);
        }
      }
      const bool need_mask = (kv0 + 64 > Sk) || (RAGGED && kv0 + 63 > q0);     // wave-uniform
      float mx = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kv0 + kt * 32 + acc_row(i, h);
            const float v = (key < Sk && (!RAGGED || key <= qi)) ? sT[kt][i] : -INFINITY;
            sT[kt][i] = v;
            mx = fmaxf(mx, v);
          }
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sT[kt][i]);
      }
      mx = fmaxf(mx, xhalf(mx));
      const float mnew = fmaxf(mrun, mx);          // finite: the first tile of every query holds >= 1 visible key
      const float alpha = exp2f((mrun - mnew) * c);       // (difference first: exact near the maximum, where p matters)
      mrun = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float p = exp2f((sT[kt][i] - mnew) * c); sT[kt][i] = p; ps += p; }
      lrun = lrun * alpha + ps;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
      // O^T[d][query] += V^T[d][key] P^T[key][query]; k-step ks covers keys 16 ks .. + 15 in the accumulator's own order: element j <-> key 16 ks + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = sT[ks >> 1][8 * (ks & 1) + j] * 32768.f; const f16 x = (f16)t; ph[j] = x; pl[j] = (f16)(t - (float)x); }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const a3_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds_s16x4)(Vh + vtr + (ks * 16) * A3_LDV + dt * 32));
          const a3_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds_s16x4)(Vh + vtr + (ks * 16 + 8) * A3_LDV + dt * 32));
          const a3_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds_s16x4)(Vl + vtr + (ks * 16) * A3_LDV + dt * 32));
          const a3_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds_s16x4)(Vl + vtr + (ks * 16 + 8) * A3_LDV + dt * 32));
          const __attribute__((ext_vector_type(8))) short vh8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          const __attribute__((ext_vector_type(8))) short vl8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
          const f16x8 vh = __builtin_bit_cast(f16x8, vh8), vl = __builtin_bit_cast(f16x8, vl8);
          o[dt] = mfma16h(vl, ph, o[dt]);
          o[dt] = mfma16h(vh, pl, o[dt]);
          o[dt] = mfma16h(vh, ph, o[dt]);
        }
      }
    }
    __syncthreads();
  }
  lrun += xhalf(lrun);
  const float inv = ldexpf(1.f, -(15 + a.v_log2)) / lrun;
  if (qvalid) {
    float* op = ob + (long long)(q0 + r) * ldo;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + 8 * q + 4 * h;
        const f32x4 v = {o[dt][4 * q] * inv, o[dt][4 * q + 1] * inv, o[dt][4 * q + 2] * inv, o[dt][4 * q + 3] * inv};
        *reinterpret_cast<f32x4*>(op + d) = v;
      }
  }
}

int launch_attn3(const Attn3Args& a, hipStream_t st) {
  const bool ragged = a.seq_row0 != nullptr;
  if (a.n_seq <= 0 || a.n_heads <= 0 || !a.Q || !a.K || !a.V || !a.O || (a.ldq % 4) || (a.ldo % 4) || (((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V | (uintptr_t)a.O) & 15))
    ETD_FAIL(ETD_EINVAL, "attn3: bad arguments");
  if (ragged && (!a.seq_len || !a.row_slot || a.max_len <= 0 || a.max_len > a.max_ctx || (a.slot_stride % 4))) ETD_FAIL(ETD_EINVAL, "attn3: bad ragged arguments");
  if (!ragged && (a.Sq <= 0 || a.Sk <= 0 || (a.ldk % 4) || (a.ldv % 4) || (a.q_seq % 4) || (a.k_seq % 4) || (a.v_seq % 4) || (a.o_seq % 4))) ETD_FAIL(ETD_EINVAL, "attn3: bad strided arguments");
  ETD_LAUNCH_FILTER("k_attn3");
  const int Sq = ragged ? a.max_len : a.Sq;
  const int nw = (Sq > 64 && Sq <= 96) ? 3 : 4;                        // 88 note queries: three waves (a tile load is sized for >= 192 threads)
  ProfScope ps("k_attn3", st, a.flops_hint, 0.0);
  const dim3 g((Sq + nw * 32 - 1) / (nw * 32), a.n_seq * a.n_heads);
  if (ragged) hipLaunchKernelGGL(k_attn3<true>, g, dim3(64 * nw), 0, st, a);
  else hipLaunchKernelGGL(k_attn3<false>, g, dim3(64 * nw), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================ test hooks (include/etude_hip_debug.h)
#include "../../include/etude_hip_debug.h"
extern "C" int etd_debug_gemm3(const float* x_dev, int M, int K, const float* w_host, const float* bias_host, int N, float x_bound, int gelu, float* y_dev,
                               const float* ln_g_host, const float* ln_b_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x_dev || !w_host || !y_dev || M < 1 || N < 1 || K < 32 || K % 32) ETD_FAIL(ETD_EINVAL, "debug_gemm3: bad arguments");
  const int Npad = (N + 127) / 128 * 128;
  std::vector<uint16_t> planes(g3_packed_elems(Npad, K));
  const int wl = g3_pack_weights_host(w_host, N, Npad, K, planes.data());
  std::vector<float> b(Npad, 0.f);
  if (bias_host) memcpy(b.data(), bias_host, (size_t)N * 4);
  uint16_t* wp = nullptr; float* bd = nullptr; float* lnd = nullptr;
  HIP_TRY(hipMalloc((void**)&wp, planes.size() * 2));
  if (hipMalloc((void**)&bd, b.size() * 4) != hipSuccess || hipMalloc((void**)&lnd, (size_t)2 * K * 4) != hipSuccess) { (void)hipFree(wp); (void)hipFree(bd); ETD_FAIL(ETD_EHIP, "debug_gemm3: hipMalloc"); }
  int rc = ETD_OK;
  if (hipMemcpy(wp, planes.data(), planes.size() * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(bd, b.data(), b.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = ETD_EHIP;
  const bool ln = ln_g_host && ln_b_host;
  if (rc == ETD_OK && ln && (hipMemcpy(lnd, ln_g_host, (size_t)K * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(lnd + K, ln_b_host, (size_t)K * 4, hipMemcpyHostToDevice) != hipSuccess)) rc = ETD_EHIP;
  if (rc == ETD_OK) {
    DGemmArgs a = {};
    a.X = x_dev; a.ldx = K; a.Wp = wp; a.w_log2 = wl; a.x_log2 = g3_scale_log2(x_bound); a.bias = bd; a.M = M; a.N = N; a.Npad = Npad; a.K = K; a.Y = y_dev; a.ldy = N;
    if (ln) { a.ln_g = lnd; a.ln_b = lnd + K; a.ln_eps = 1e-5f; }       // (x_bound then bounds the LayerNorm OUTPUT; the fused LayerNorm exists in the small-M kernel only)
    const int epi = gelu ? DEPI_GELU : DEPI_BIAS;
    rc = gemm3_s_takes(a, epi) ? launch_gemm3_s(a, epi, st) : launch_gemm3(a, epi, st);
  }
  if (hipStreamSynchronize(st) != hipSuccess && rc == ETD_OK) { g_etd_err = "debug_gemm3: kernel failed"; rc = ETD_EHIP; }
  (void)hipFree(wp); (void)hipFree(bd); (void)hipFree(lnd);
  return rc;
}
extern "C" int etd_debug_attn3(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int n_seq, int n_heads, int Sq, int Sk, float q_bound, float k_bound, float v_bound,
                               int causal, const int32_t* lens_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!q_dev || !k_dev || !v_dev || !o_dev || n_seq < 1 || n_heads < 1 || Sq < 1 || Sk < 1) ETD_FAIL(ETD_EINVAL, "debug_attn3: bad arguments");
  const int H = n_heads * 64;
  Attn3Args a = {};
  a.Q = q_dev; a.ldq = H; a.K = k_dev; a.V = v_dev; a.O = o_dev; a.ldo = H; a.n_seq = n_seq; a.n_heads = n_heads; a.scale = 0.125f;
  a.q_log2 = g3_scale_log2(q_bound); a.k_log2 = g3_scale_log2(k_bound); a.v_log2 = g3_scale_log2(v_bound);
  int* meta = nullptr;
  int rc = ETD_OK;
  if (causal) {
    if (!lens_host || Sq != Sk) ETD_FAIL(ETD_EINVAL, "debug_attn3: causal needs lens_host and Sq == Sk");
    std::vector<int> h;                              // [seq_row0 n][seq_len n][row_slot M]
    int M = 0, mx = 0;
    for (int s = 0; s < n_seq; ++s) { if (lens_host[s] < 1 || lens_host[s] > Sq) ETD_FAIL(ETD_EINVAL, "debug_attn3: bad prompt length"); M += lens_host[s]; mx = lens_host[s] > mx ? lens_host[s] : mx; }
    h.resize((size_t)2 * n_seq + M);
    int row = 0;
    for (int s = 0; s < n_seq; ++s) { h[s] = row; h[n_seq + s] = lens_host[s]; for (int t = 0; t < lens_host[s]; ++t) h[2 * n_seq + row++] = s; }
    HIP_TRY(hipMalloc((void**)&meta, h.size() * 4));
    if (hipMemcpy(meta, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = ETD_EHIP;
    a.seq_row0 = meta; a.seq_len = meta + n_seq; a.row_slot = meta + 2 * n_seq; a.slot_stride = (long long)n_heads * Sk * 64; a.max_ctx = Sk; a.max_len = mx;
  } else {
    a.ldk = H; a.ldv = H; a.q_seq = (long long)Sq * H; a.k_seq = (long long)Sk * H; a.v_seq = (long long)Sk * H; a.o_seq = (long long)Sq * H; a.Sq = Sq; a.Sk = Sk;
  }
  if (rc == ETD_OK) rc = launch_attn3(a, st);
  if (hipStreamSynchronize(st) != hipSuccess && rc == ETD_OK) { g_etd_err = "debug_attn3: kernel failed"; rc = ETD_EHIP; }
  if (meta) (void)hipFree(meta);
  return rc;
}

// host-only test hook: the load-time bounds the plane scales come from (tests/test_host_logic.py checks that they bound)
extern "C" int etd_debug_g3_bounds(const float* W, const float* c, int N, int K, const float* g, const float* b, float elem_bound, float* out4, int32_t* log2_out4) {
  if (!W || !g || !b || !out4 || !log2_out4 || N < 1 || K < 1) ETD_FAIL(ETD_EINVAL, "debug_g3_bounds: bad arguments");
  out4[0] = g3_bound_ln(g, b, K);
  out4[1] = g3_bound_linear_of_ln(W, c, N, K, g, b);
  out4[2] = g3_bound_linear(W, c, N, K, elem_bound);
  std::vector<uint16_t> planes(g3_packed_elems((N + 127) / 128 * 128, (K + 31) / 32 * 32));
  out4[3] = 0.f;
  for (int i = 0; i < 3; ++i) log2_out4[i] = g3_scale_log2(out4[i]);
  log2_out4[3] = (K % 32 == 0) ? g3_pack_weights_host(W, N, (N + 127) / 128 * 128, K, planes.data()) : 0;
  if (K % 32 == 0) {        // largest plane magnitude actually stored: must sit in [2^14, 2^15)
    float mx = 0.f;
    for (size_t i = 0; i < planes.size(); ++i) { f16 h; memcpy(&h, &planes[i], 2); const float v = fabsf((float)h); mx = v > mx ? v : mx; }
    out4[3] = mx;
  }
  return ETD_OK;
}
