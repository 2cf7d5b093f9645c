// EtudeDecoder (GPT-NeoX, 8 x [LN, QKV+RoPE, causal attention, dense | LN, MLP(GELU)] parallel residual)
// kernels for gfx950.  Reference: etude/models/etude_decoder.py:148-206 and HF GPTNeoXLayer /
// GPTNeoXAttention (transformers modeling_gpt_neox.py:195-281, RoPE :111-151).
//
// Two weight precisions share every kernel:
//   * fp32 ("parity mode"): exact-fp32 products on v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf
//     chain), fp32 KV cache -- this is what the greedy token-id parity gate runs on;
//   * d16: v_mfma_f32_32x32x16_bf16 on d16 weights and a d16 KV cache (the HBM-bound serving mode).
// The residual stream, LayerNorm, RoPE, softmax and the logits are fp32 in both.
#include "dec_kernels.h"
#include "prof.h"

#define DLD32 36   // fp32 LDS row stride (floats) for a 32-wide K chunk: 144 B
#define DLD16 72   // d16 LDS row stride (elements) for a 64-wide K chunk: 144 B

#include "dec_epilogue.h"

// Diagnostic build (-DETD_STEP_STAMP, tools/step_stamps.py): every workgroup of the three per-layer decode-step kernels records
// s_memrealtime (100 MHz, one clock for the whole chip) at its phase boundaries into a device buffer handed over with
// etd_debug_step_stamps.  Record = 8 x int64: [kernel id | role << 8 | blockIdx << 16, t0 .. t5, XCC id].  The shipped build has none.
#ifdef ETD_STEP_STAMP
__device__ long long* g_ss_buf;
__device__ unsigned long long g_ss_cap;
__device__ unsigned long long g_ss_cnt;
#define SS_DECL() long long ss_t[6] = {0, 0, 0, 0, 0, 0}
#define SS(i) do { __builtin_amdgcn_sched_barrier(0); ss_t[i] = (long long)__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SS_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SS_FLUSH(kid, role) do { if (threadIdx.x == 0 && g_ss_buf) { const unsigned long long i_ = atomicAdd(&g_ss_cnt, 1ull); if (i_ < g_ss_cap) { long long* o_ = g_ss_buf + i_ * 8;          \
      o_[0] = (long long)(kid) | ((long long)(role) << 8) | ((long long)blockIdx.x << 16); o_[1] = ss_t[0]; o_[2] = ss_t[1]; o_[3] = ss_t[2]; o_[4] = ss_t[3]; o_[5] = ss_t[4]; o_[6] = ss_t[5];   \
      o_[7] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); } } } while (0)      /* HW_REG_XCC_ID bits [3:0] */
extern "C" int etd_debug_step_stamps(long long* dev_buf, unsigned long long cap_records, unsigned long long* count_out) {
  HIP_TRY(hipDeviceSynchronize());
  if (count_out) HIP_TRY(hipMemcpyFromSymbol(count_out, HIP_SYMBOL(g_ss_cnt), 8));
  const unsigned long long zero = 0;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ss_cnt), &zero, 8));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ss_buf), &dev_buf, 8));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_ss_cap), &cap_records, 8));
  return ETD_OK;
}
#else
#define SS_DECL() do { } while (0)
#define SS(i) do { } while (0)
#define SS_LANDED() do { } while (0)
#define SS_FLUSH(kid, role) do { } while (0)
#endif

// ================================================================================================
// k_dgemm: Y[M,N] = epi( LN?(X)[M,K] * W[N,K]^T + b ).  Workgroup = 4 waves = 32 tokens x 128
// features (each wave one 32x32 accumulator, token on the lane).
// ================================================================================================
template <bool WBF16, int EPI>
__global__ __launch_bounds__(256) void k_dgemm(DGemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(32 + 128) * 144 + 64 * 4];
  float* stat = reinterpret_cast<float*>(smem + (32 + 128) * 144);      // mean[32] | rstd[32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 128;
  const bool ln = a.ln_g != nullptr;

  if (ln) {
    // each wave: 8 rows; LayerNorm statistics over K (= hidden, multiple of 256)
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
      const float* xp = a.X + (long long)gm * a.ldx;
      float s = 0.f;
      for (int k = lane * 4; k < a.K; k += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k); s += v[0] + v[1] + v[2] + v[3]; }
      s = wave_sum(s);
      const float mean = s / (float)a.K;
      float q = 0.f;
      for (int k = lane * 4; k < a.K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k);
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
      q = wave_sum(q);
      if (lane == 0) { stat[row] = mean; stat[32 + row] = rsqrtf(q / (float)a.K + a.ln_eps); }
    }
    __syncthreads();
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  if constexpr (!WBF16) {
    float* Xs = reinterpret_cast<float*>(smem);
    float* Ws = Xs + 32 * DLD32;
    const float* W = reinterpret_cast<const float*>(a.W) + (long long)n0 * a.K;
    for (int k0 = 0; k0 < a.K; k0 += 32) {
      {  // X: 32 rows x 8 float4
        const int row = tid >> 3, ch = tid & 7;
        int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.X + (long long)gm * a.ldx + k0 + ch * 4);
        if (ln) {
          const float mean = stat[row], rstd = stat[32 + row];
          const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g + k0 + ch * 4);
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.ln_b + k0 + ch * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) * rstd * g[j] + b[j];
        }
        *reinterpret_cast<f32x4*>(Xs + row * DLD32 + ch * 4) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // W: 128 rows x 8 float4
        const int c = tid + i * 256, row = c >> 3, ch = c & 7;
        *reinterpret_cast<f32x4*>(Ws + row * DLD32 + ch * 4) = *reinterpret_cast<const f32x4*>(W + (long long)row * a.K + k0 + ch * 4);
      }
      __syncthreads();
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const f32x4 wf = *reinterpret_cast<const f32x4*>(Ws + (wave * 32 + r) * DLD32 + jp * 8 + h * 4);
        const f32x4 xf = *reinterpret_cast<const f32x4*>(Xs + r * DLD32 + jp * 8 + h * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s], xf[s], acc, 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
    d16* Xs = reinterpret_cast<d16*>(smem);
    d16* Ws = Xs + 32 * DLD16;
    const d16* W = reinterpret_cast<const d16*>(a.W) + (long long)n0 * a.K;
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      {  // X: 32 rows x 8 chunks of 8 floats -> d16
        const int row = tid >> 3, ch = tid & 7;
        int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
        const float* xp = a.X + (long long)gm * a.ldx + k0 + ch * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
        if (ln) {
          const float mean = stat[row], rstd = stat[32 + row];
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + k0 + ch * 8), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + k0 + ch * 8 + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + k0 + ch * 8), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + k0 + ch * 8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v0[j] = (v0[j] - mean) * rstd * g0[j] + b0[j]; v1[j] = (v1[j] - mean) * rstd * g1[j] + b1[j]; }
        }
        d16x8 o = {(d16)v0[0], (d16)v0[1], (d16)v0[2], (d16)v0[3], (d16)v1[0], (d16)v1[1], (d16)v1[2], (d16)v1[3]};
        *reinterpret_cast<d16x8*>(Xs + row * DLD16 + ch * 8) = o;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // W: 128 rows x 8 chunks of 16 B
        const int c = tid + i * 256, row = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(Ws + row * DLD16 + ch * 8) = *reinterpret_cast<const u32x4*>(W + (long long)row * a.K + k0 + ch * 8);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const d16x8 wf = *reinterpret_cast<const d16x8*>(Ws + (wave * 32 + r) * DLD16 + s * 16 + h * 8);
        const d16x8 xf = *reinterpret_cast<const d16x8*>(Xs + r * DLD16 + s * 16 + h * 8);
        acc = mfma32(wf, xf, acc);
      }
      __syncthreads();
    }
  }

  const int m = m0 + r;
  if (m >= a.M) return;
  dgemm_epilogue<WBF16, EPI>(a, acc, m, n0 + wave * 32, h);
}

// ================================================================================================
// k_dgemm_s ("skinny", M <= DS_MAX_ROWS): 32 tokens x 32 features per workgroup, K split over the 4 waves,
// fragments straight from global/L2 to registers (no LDS staging, no barrier in the K loop), partial
// tiles reduced through LDS in a fixed order (bit-reproducible).  (N/32) x (M/32) workgroups keep many
// more CUs streaming weights than the 128-feature tile does when M is a handful of decode rows.
// ================================================================================================
// (DS_MAX_ROWS / DS_STEP_MAX_ROWS: dec_kernels.h)
#ifndef DS_WAVES
#define DS_WAVES 2
#endif
// K is split over the waves of a workgroup; each wave keeps 512 / DS_WAVES of k (32 fragment loads) in flight per round trip.
                     // Measured per 128-stream step: 8 waves 0.423 ms, 4 waves 0.404 ms, 2 waves 0.385 ms (fewer wave slots held, shorter LDS reduction)
template <bool WBF16, int EPI>
__global__ __launch_bounds__(64 * DS_WAVES) void k_dgemm_s(int p_M, int p_Npad, int p_ks, int p_K, const void* p_W, const d16* p_Xb, int p_ldx, DGemmArgs a) {   // leading scalars: kernarg preload
  __shared__ __attribute__((aligned(16))) float red[DS_WAVES - 1][16][64];
  __shared__ float stat[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  // Workgroup -> tile: the row tiles that stream the SAME weight tile (feature block x K slice) take consecutive slots of
  // ONE XCD (id = 8*slot + xcd), so the weights come from HBM / the Infinity Cache once and from that XCD's L2 afterwards
  // (FETCH_SIZE before: 12.9 MB per QKV|up launch at 72 rows for 3.7 MB of weights).
  const int RT = (p_M + 31) / 32, FT = p_Npad / 32, KS = p_ks > 1 ? p_ks : 1;
  const int bid = blockIdx.x, wt = ((bid >> 3) / RT) * 8 + (bid & 7);
  if (wt >= FT * KS) return;
  const int bz = wt / FT;
  const int m0 = ((bid >> 3) % RT) * 32, n0 = (wt - bz * FT) * 32;
  const bool ln = a.ln_g != nullptr;
  if (ln) {
    for (int rr = 0; rr < 32 / DS_WAVES; ++rr) {
      const int row = wave * (32 / DS_WAVES) + rr;
      int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
      const float* xp = a.X + (long long)gm * a.ldx;
      float s = 0.f;
      for (int k = lane * 4; k < a.K; k += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k); s += v[0] + v[1] + v[2] + v[3]; }
      s = wave_sum(s);
      const float mean = s / (float)a.K;
      float q = 0.f;
      for (int k = lane * 4; k < a.K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + k);
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
      q = wave_sum(q);
      if (lane == 0) { stat[row] = mean; stat[32 + row] = rsqrtf(q / (float)a.K + a.ln_eps); }
    }
    __syncthreads();
  }
  int gm = m0 + r; gm = gm < p_M ? gm : p_M - 1;
  const float mean = ln ? stat[r] : 0.f, rstd = ln ? stat[32 + r] : 1.f;
  const float* xrow = a.X + (long long)gm * a.ldx;
  const int Kz = p_K / KS;                             // split-K over workgroups (DEPI_PARTIAL), then over the waves
  const int kq = Kz / DS_WAVES, kb = bz * Kz + wave * kq, ke = kb + kq;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if constexpr (WBF16) {
    const d16* wrow = reinterpret_cast<const d16*>(p_W) + (long long)(n0 + r) * p_K;
    if (p_Xb) {
      const d16* xbrow = p_Xb + (long long)gm * p_ldx;
      int k = kb;
      constexpr int PW = 512 / DS_WAVES, PN = PW / 16;      // k covered per round trip by one wave: all its fragment loads in flight, then the MFMAs
      for (; k + PW <= ke; k += PW) {
        d16x8 wf[PN], xf[PN];
#pragma unroll
        for (int s8 = 0; s8 < PN; ++s8) {
          wf[s8] = *reinterpret_cast<const d16x8*>(wrow + k + s8 * 16 + h * 8);
          xf[s8] = *reinterpret_cast<const d16x8*>(xbrow + k + s8 * 16 + h * 8);
        }
        __builtin_amdgcn_sched_barrier(0);     // every load of the pass is issued before the first MFMA waits: ONE round trip
#pragma unroll
        for (int s8 = 0; s8 < PN; ++s8) acc = mfma32(wf[s8], xf[s8], acc);
      }
      for (; k < ke; k += 64) {
        d16x8 wf[4], xf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          wf[s4] = *reinterpret_cast<const d16x8*>(wrow + k + s4 * 16 + h * 8);
          xf[s4] = *reinterpret_cast<const d16x8*>(xbrow + k + s4 * 16 + h * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = mfma32(wf[s4], xf[s4], acc);
      }
    } else {
#pragma unroll 2
      for (int k = kb; k < ke; k += 64) {
        d16x8 wf[4]; f32x4 x0[4], x1[4];
  #pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          wf[s4] = *reinterpret_cast<const d16x8*>(wrow + k + s4 * 16 + h * 8);
          x0[s4] = *reinterpret_cast<const f32x4*>(xrow + k + s4 * 16 + h * 8);
          x1[s4] = *reinterpret_cast<const f32x4*>(xrow + k + s4 * 16 + h * 8 + 4);
        }
  #pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          if (ln) {
            const int kk = k + s4 * 16 + h * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + kk), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + kk + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + kk), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + kk + 4);
  #pragma unroll
            for (int j = 0; j < 4; ++j) { x0[s4][j] = (x0[s4][j] - mean) * rstd * g0[j] + b0[j]; x1[s4][j] = (x1[s4][j] - mean) * rstd * g1[j] + b1[j]; }
          }
          const d16x8 xf = {(d16)x0[s4][0], (d16)x0[s4][1], (d16)x0[s4][2], (d16)x0[s4][3], (d16)x1[s4][0], (d16)x1[s4][1], (d16)x1[s4][2], (d16)x1[s4][3]};
          acc = mfma32(wf[s4], xf, acc);
        }
      }
    }
  } else {
    const float* wrow = reinterpret_cast<const float*>(a.W) + (long long)(n0 + r) * a.K;
#pragma unroll 2
    for (int k = kb; k < ke; k += 32) {
      f32x4 wf[4], xf[4];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        wf[jp] = *reinterpret_cast<const f32x4*>(wrow + k + jp * 8 + h * 4);
        xf[jp] = *reinterpret_cast<const f32x4*>(xrow + k + jp * 8 + h * 4);
      }
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        if (ln) {
          const int kk = k + jp * 8 + h * 4;
          const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g + kk), b = *reinterpret_cast<const f32x4*>(a.ln_b + kk);
#pragma unroll
          for (int j = 0; j < 4; ++j) xf[jp][j] = (xf[jp][j] - mean) * rstd * g[j] + b[j];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[jp][s4], xf[jp][s4], acc, 0, 0, 0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < DS_WAVES - 1; ++w)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
  const int m = m0 + r;
  if (m >= a.M) return;
  if constexpr (EPI == DEPI_PARTIAL) {
    DGemmArgs b = a;
    b.Y = a.Y + (long long)bz * a.M * a.ldy;
    dgemm_epilogue<WBF16, EPI>(b, acc, m, n0, h);
  } else {
    dgemm_epilogue<WBF16, EPI>(a, acc, m, n0, h);
  }
}

// ================================================================================================
// k_dstep_qkv_up: the two GEMMs of a decode step that only depend on the layer's LayerNorm rows -- the fused QKV
// projection (X1b, +RoPE, KV append) and the MLP up projection (X2b, +GELU) -- in ONE launch: feature tiles
// [0, split) belong to the first problem, the rest to the second.  Same 32x32 / 8-wave K-split body as k_dgemm_s
// (d16 inputs, K = hidden).  One dependent kernel boundary less per layer of a latency-bound chain.
// ================================================================================================
// Epilogue of one 32-feature x 32-token tile of the decode step's QKV|up GEMM: lane = token m, register i = feature n0 + acc_row(i, h).
// QKV tiles ([head][q|k|v][64], modeling_gpt_neox.py:204-207): bias, partial RoPE on the first 16 dims of q / k, Q as fp32 rows, K / V appended
// to the slot's cache rows; up tiles: bias + erf-GELU -> d16 rows of Xcat.  Shared by k_dstep_qkv_up and k_dstep_qkv_up_mt (same operations in the
// same order: bit-identical results).
__device__ __forceinline__ void qkv_up_tile_values(const f32x16& acc, const f32x4 (&bq)[4], bool isq, bool rope, const f32x4& rc, const f32x4& rs, float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = acc[i] + bq[i >> 2][i & 3];
  if (!isq) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = gelu_fast(v[i]);     // MLP up: erf-GELU
    return;
  }
  if (rope) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x1 = v[i], x2 = v[i + 4];
      v[i] = x1 * rc[i] - x2 * rs[i];        // q*cos + rotate_half(q)*sin, first half
      v[i + 4] = x2 * rc[i] + x1 * rs[i];    // second half
    }
  }
}
// (k_dstep_qkv_up's own form: the same operations, each group of four values rounded and stored as soon as it is formed -- forming all sixteen first, as
// qkv_up_tile_values does for the LDS-staged stores of the mid-tile kernel, costs this kernel 30 registers and its second wave per SIMD)
__device__ __forceinline__ void qkv_up_store_tile(const f32x16& acc, const f32x4 (&bq)[4], bool isq, bool rope, const f32x4& rc, const f32x4& rs, int part, int head, int dbase,
                                                  int n0, int m, int h, int pos, int slot, int act, const DGemmArgs& q, const DGemmArgs& up) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = acc[i] + bq[i >> 2][i & 3];
  if (!isq) {
    // MLP up: erf-GELU, d16 rows for the down projection
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int n = n0 + 8 * q4 + 4 * h;
      if (n < up.N)
        *reinterpret_cast<d16x4*>(up.Yb + (long long)m * up.ldy + n) =
            pack4d(gelu_fast(v[4 * q4]), gelu_fast(v[4 * q4 + 1]), gelu_fast(v[4 * q4 + 2]), gelu_fast(v[4 * q4 + 3]));
    }
    return;
  }
  if (rope) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x1 = v[i], x2 = v[i + 4];
      v[i] = x1 * rc[i] - x2 * rs[i];        // q*cos + rotate_half(q)*sin, first half
      v[i + 4] = x2 * rc[i] + x1 * rs[i];    // second half
    }
  }
  if (part == 0) {
    float* qp = q.Q + (long long)m * (q.n_heads * 64) + head * 64 + dbase;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { const f32x4 o = {v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]}; *reinterpret_cast<f32x4*>(qp + 8 * q4 + 4 * h) = o; }
  } else if (act && pos < q.max_ctx) {
    d16* kp = reinterpret_cast<d16*>(part == 1 ? q.Kc : q.Vc) + (long long)slot * q.slot_stride + ((long long)head * q.max_ctx + pos) * 64 + dbase;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<d16x4*>(kp + 8 * q4 + 4 * h) = pack4d(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
  }
}

__global__ __launch_bounds__(64 * DS_WAVES) void k_dstep_qkv_up(int p_M, int p_K, int split, int p_ftiles, const void* p_Wq, const void* p_Wu, const d16* p_Xq,
                                                               const d16* p_Xu, int p_ldxq, int p_ldxu, int p_rpt, DGemmArgs q, DGemmArgs up) {   // leading scalars: kernarg preload
  __shared__ __attribute__((aligned(16))) float red[DS_WAVES - 1][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int M = p_M, K = p_K;
  SS_DECL(); SS(0);
  // workgroup -> (feature tile, group of p_rpt consecutive row tiles) as in k_dgemm_s: the row tiles of one weight tile run back to back on one XCD.
  // p_rpt = 1 (up to 512 rows: the latency-bound regime wants every tile on its own workgroup) or 4 (above: a workgroup keeps its weight
  // fragments in registers for four row tiles -- a quarter of the weight traffic from L2 and of the workgroups; round 3: 50 -> ~20 us at 1728 rows).
  // Per row tile the same loads, MFMAs and epilogue in the same order: bit-identical to p_rpt = 1.
  const int RT = (M + 31) / 32, RG = (RT + p_rpt - 1) / p_rpt, bid = blockIdx.x, ft = ((bid >> 3) / RG) * 8 + (bid & 7);
  if (ft >= p_ftiles) return;
  const bool isq = ft < split;
  const int rt0 = ((bid >> 3) % RG) * p_rpt, n0 = (isq ? ft : ft - split) * 32;
  // fused QKV laid out [head][q|k|v][64] (modeling_gpt_neox.py:204-207); a 32-feature tile is half of one part of one head
  const int head = n0 / 192, j0 = n0 - head * 192, part = j0 >> 6, dbase = j0 & 63;
  const bool rope = isq && part < 2 && dbase == 0;
  // Wave 0 owns the epilogue.  Its row metadata and bias are requested BEFORE the weight stream and the RoPE factors
  // (which depend on the row's position) right after it, so these dependent round trips overlap the K loop and the LDS
  // reduction instead of following them.
  // (Every wave issues these few loads, unconditionally: putting them under `if (wave == 0)` / `if (isq)` makes the compiler merge
  // the loaded values with the defaults right behind the branch, i.e. wait a full round trip BEFORE the weight stream is issued.)
  f32x4 bq[4];
  int gm = rt0 * 32 + r; gm = gm < M ? gm : M - 1;
  int pos = q.rows.pos[gm], slot = q.rows.slot[gm], act = q.rows.active[gm];
  {
    const float* bp = (isq ? q.bias : up.bias) + n0 + 4 * h;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) bq[q4] = *reinterpret_cast<const f32x4*>(bp + 8 * q4);
  }
  const d16* wrow = reinterpret_cast<const d16*>(isq ? p_Wq : p_Wu) + (long long)(n0 + r) * K;
  const d16* xbase = isq ? p_Xq : p_Xu;
  const int ldx = isq ? p_ldxq : p_ldxu;
  const int kq = K / DS_WAVES, kb = wave * kq;
  constexpr int NS = 8 / DS_WAVES * 4;           // 16-wide k-steps per wave at K = 512 (launcher-checked: K == 64 * DS_WAVES * NS / 4)
  d16x8 wf[NS], xf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    wf[s] = *reinterpret_cast<const d16x8*>(wrow + kb + s * 16 + h * 8);
    xf[s] = *reinterpret_cast<const d16x8*>(xbase + (long long)gm * ldx + kb + s * 16 + h * 8);
  }
  for (int t = 0; t < p_rpt; ++t) {
    const int m0 = (rt0 + t) * 32;
    if (m0 >= M) break;                              // (uniform over the workgroup)
    f32x4 rc = {1.f, 1.f, 1.f, 1.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && rope) {
      // partial RoPE factors of dims d = 4h + i (i < 4); the partner d + 8 sits in register i + 4 of the same lane
      rc = *reinterpret_cast<const f32x4*>(q.rope_cos + (long long)pos * 8 + 4 * h);
      rs = *reinterpret_cast<const f32x4*>(q.rope_sin + (long long)pos * 8 + 4 * h);
    }
    __builtin_amdgcn_sched_barrier(0);
    SS_LANDED(); SS(1);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = mfma32(wf[s], xf[s], acc);
    // the next row tile's fragments and row metadata are requested here: their round trip runs under this tile's reduction and epilogue
    const int posc = pos, slotc = slot, actc = act;
    if (t + 1 < p_rpt && m0 + 32 < M) {
      int gn = m0 + 32 + r; gn = gn < M ? gn : M - 1;
#pragma unroll
      for (int s = 0; s < NS; ++s) xf[s] = *reinterpret_cast<const d16x8*>(xbase + (long long)gn * ldx + kb + s * 16 + h * 8);
      pos = q.rows.pos[gn]; slot = q.rows.slot[gn]; act = q.rows.active[gn];
    }
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
      SS(2);
#pragma unroll
      for (int w = 0; w < DS_WAVES - 1; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
      const int m = m0 + r;
      if (m < M) qkv_up_store_tile(acc, bq, isq, rope, rc, rs, part, head, dbase, n0, m, h, posc, slotc, actc, q, up);
    }
    if (t + 1 < p_rpt) __syncthreads();              // wave 0 has read `red` before the next tile's partial sums land in it
  }
  SS(3); SS_FLUSH(1, isq ? 0 : 1);
}

// k_dstep_qkv_up_mt: the same GEMM for MANY rows (above DS_MAX_ROWS: 1 728 streams of a 64-clip batch in one launch).  What the 32 x 32 form costs there
// (38-41 us at 1 728 rows, whether or not a workgroup keeps its weights for several row tiles) is neither its operand bytes from L2 nor its stores: a fragment load
// straight from a ROW-MAJOR operand touches 32 rows x 32 bytes -- 32 cache lines per instruction, a quarter of each used -- and the CU's address path, not the L2,
// sets the pace (a 128 x 128 tile fed the same way took the same 39 us; so did LDS-staged row-segment stores alone).  Here a workgroup owns 128 features x 128
// rows, each of its 4 waves 64 x 64 (2 x 2 MFMA tiles):
//   * weights from the FRAGMENT-ORDERED copy the batched prefill already uses (pack_wfrag_host: one contiguous KiB per fragment, lane l's 16 bytes at 16 l);
//   * activations through LDS: a 64-deep K chunk of the 128 rows arrives as full 128-byte row segments (8 lanes per row), double-buffered, fragments by
//     conflict-free ds_read_b128 from 144-byte rows;
//   * outputs leave as full row segments through the same LDS (below).
// Bit-identical to the 32 x 32 form: a tile's sum is still (k 0 .. 255 chained) + (k 256 .. 511 chained), as that kernel's two K-half waves form it.
#define QMT_PITCH 144                                    // bytes per LDS row of 64 d16 (+16: conflict-free 16-byte fragment reads)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dstep_qkv_up_mt(int p_M, int split128, int p_ftiles128, const d16* p_Wfq, const d16* p_Wfu, const d16* p_Xq, const d16* p_Xu,
                                                         int p_ldxq, int p_ldxu, DGemmArgs q, DGemmArgs up) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * QMT_PITCH];      // K loop: two [128 rows][64 k] chunks of X; epilogue: four [64 rows][64 features] wave regions
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5, fh = wave & 1, rh = wave >> 1;
  const int M = p_M, K = 512;
  const int RT = (M + 127) / 128, bid = blockIdx.x, ft = ((bid >> 3) / RT) * 8 + (bid & 7);      // row tiles of one weight tile back to back on one XCD
  if (ft >= p_ftiles128) return;
  const bool isq = ft < split128;
  const int mwg = ((bid >> 3) % RT) * 128, m0 = mwg + rh * 64, n0 = (isq ? ft : ft - split128) * 128 + fh * 64;
  const d16* wfrag = (isq ? p_Wfq : p_Wfu) + ((long long)(n0 / 32) * (K / 16) * 64 + lane) * 8;      // fragment (tile n0 / 32 + t, k-step s) at + ((t * 32 + s) * 64) * 8
  const d16* xbase = isq ? p_Xq : p_Xu;
  const int ldx = isq ? p_ldxq : p_ldxu;
  // this thread's four 16-byte pieces of a chunk: piece p = tid + 256 i -> row p >> 3, 16-byte column p & 7
  const d16* xsrc[4]; int xdst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pce = tid + 256 * i, row = pce >> 3, c = pce & 7;
    int gmr = mwg + row; gmr = gmr < M ? gmr : M - 1;
    xsrc[i] = xbase + (long long)gmr * ldx + c * 8;
    xdst[i] = row * QMT_PITCH + c * 16;
  }
  int gm[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { gm[t] = m0 + 32 * t + r; gm[t] = gm[t] < M ? gm[t] : M - 1; }
  // row metadata and bias first (the epilogue's dependent loads ride under the K loop)
  int pos[2], slot[2], act[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { pos[t] = q.rows.pos[gm[t]]; slot[t] = q.rows.slot[gm[t]]; act[t] = q.rows.active[gm[t]]; }
  f32x4 bq[2][4];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const float* bp = (isq ? q.bias : up.bias) + n0 + 32 * f + 4 * h;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) bq[f][q4] = *reinterpret_cast<const f32x4*>(bp + 8 * q4);
  }
  f32x16 acc[2][2], accA[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[f][t][i] = 0.f;
  u32x4 xr[4];                                           // the next chunk's pieces on their way to LDS
  d16x8 wf[2][4][2];                                    // [buffer][k-step of the chunk][feature tile]
  auto xrequest = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + c * 64);
  };
  auto wrequest = [&](int c, int b) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int f = 0; f < 2; ++f) wf[b][s4][f] = *reinterpret_cast<const d16x8*>(wfrag + (long long)((f * 32 + c * 4 + s4) * 64) * 8);
  };
  xrequest(0); wrequest(0, 0);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = half * 4 + cc, b = c & 1;
      unsigned char* xb = smem + b * (128 * QMT_PITCH);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(xb + xdst[i]) = xr[i];
      __syncthreads();                                   // chunk c is in LDS; every wave is done reading chunk c - 1's buffer twin (c - 2 used this one)
      if (c + 1 < 8) { xrequest(c + 1); wrequest(c + 1, b ^ 1); }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        d16x8 xf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) xf[t] = *reinterpret_cast<const d16x8*>(xb + (rh * 64 + 32 * t + r) * QMT_PITCH + s4 * 32 + h * 16);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[f][t] = mfma32(wf[b][s4][f], xf[t], acc[f][t]);
      }
    }
    if (half == 0) {                                     // k 0 .. 255 done: what the 32 x 32 kernel's first K-half wave holds
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          accA[f][t] = acc[f][t];
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[f][t][i] = 0.f;
        }
    }
  }
  __syncthreads();                                       // the X buffers become the waves' output staging regions
  // ---- epilogue.  With the token on the lane a direct store puts 8 (d16) or 16 (fp32) bytes into each of 32 different rows per instruction: 1.5 M partial
  // writes per 1 728-row launch, and THAT, not the operand traffic, set the 39 us of both tile shapes.  The wave's 64 features are exactly one (head, part) of the
  // fused QKV -- one 128-byte K / V cache row, one 256-byte run of a Q row -- or 128 bytes of an Xcat row: the values go through the wave's own LDS region
  // ([64 rows][64 features], row pitch + 16 B: conflict-free) and leave as full row segments, 8 lanes per 128 bytes.
  unsigned char* sw = smem + wave * (64 * QMT_PITCH);
  const int head = n0 / 192, part = (n0 - head * 192) >> 6;       // (QKV tiles; n0 is a multiple of 64)
  const int er = lane >> 3, ec = lane & 7;                        // read-back: 8 rows x 8 sixteen-byte chunks per instruction
  float vv[2][2][16];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const bool rope = isq && part < 2 && f == 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 a2;
#pragma unroll
      for (int i = 0; i < 16; ++i) a2[i] = accA[f][t][i] + acc[f][t][i];      // (first half) + (second half): the order of the two-wave reduction
      f32x4 rc = {1.f, 1.f, 1.f, 1.f}, rs = {0.f, 0.f, 0.f, 0.f};
      if (rope) {
        rc = *reinterpret_cast<const f32x4*>(q.rope_cos + (long long)pos[t] * 8 + 4 * h);
        rs = *reinterpret_cast<const f32x4*>(q.rope_sin + (long long)pos[t] * 8 + 4 * h);
      }
      qkv_up_tile_values(a2, bq[f], isq, rope, rc, rs, vv[f][t]);
    }
  }
  if (!isq || part != 0) {
    // d16 rows: Xcat (up) or the K / V cache row of (slot, head, position)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *reinterpret_cast<d16x4*>(sw + (32 * t + r) * 144 + (32 * f + 8 * q4 + 4 * h) * 2) = pack4d(vv[f][t][4 * q4], vv[f][t][4 * q4 + 1], vv[f][t][4 * q4 + 2], vv[f][t][4 * q4 + 3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + er, m = m0 + row;
      const u32x4 val = *reinterpret_cast<const u32x4*>(sw + row * 144 + ec * 16);
      // the row's metadata sits in lane (row & 31) of the wave: fetched with EVERY lane active (a shuffle from a lane that has left the loop is undefined --
      // a ragged last tile, e.g. a 300-row prefill, would append K / V rows at garbage positions)
      const int rl = row & 31, tt = row >> 5;
      const int ps = __shfl(tt ? pos[1] : pos[0], rl, 64), sl = __shfl(tt ? slot[1] : slot[0], rl, 64), ac = __shfl(tt ? act[1] : act[0], rl, 64);
      if (m >= M) continue;
      if (!isq) {
        if (n0 + ec * 8 < up.N) *reinterpret_cast<u32x4*>(up.Yb + (long long)m * up.ldy + n0 + ec * 8) = val;
      } else {
        if (ac && ps < q.max_ctx)
          *reinterpret_cast<u32x4*>(reinterpret_cast<d16*>(part == 1 ? q.Kc : q.Vc) + (long long)sl * q.slot_stride + ((long long)head * q.max_ctx + ps) * 64 + ec * 8) = val;
      }
    }
  } else {
    // Q: fp32 rows, 256 bytes per (row, head): two passes of 32 features through the same region
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 o = {vv[f][t][4 * q4], vv[f][t][4 * q4 + 1], vv[f][t][4 * q4 + 2], vv[f][t][4 * q4 + 3]};
          *reinterpret_cast<f32x4*>(sw + (32 * t + r) * 144 + (8 * q4 + 4 * h) * 4) = o;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + er, m = m0 + row;
        const u32x4 val = *reinterpret_cast<const u32x4*>(sw + row * 144 + ec * 16);
        if (m < M) *reinterpret_cast<u32x4*>(q.Q + (long long)m * (q.n_heads * 64) + head * 64 + 32 * f + ec * 4) = val;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

int launch_dstep_qkv_up(const DGemmArgs& q, const DGemmArgs& up, hipStream_t st) {
  if (q.M != up.M || q.K != up.K || q.M < 1 || q.M > DS_STEP_MAX_ROWS || !q.Xb || !up.Xb || !up.Yb || q.K != 512 || q.Npad % 32 || up.Npad % 32 ||
      q.rot_half != 8 || q.N % 192 || q.Qb || !q.Q || !q.bias || !up.bias || up.N % 4)
    ETD_FAIL(ETD_EINVAL, "dstep_qkv_up: bad arguments");
  ProfScope ps("k_dstep_qkv_up", st, 2.0 * q.M * (q.N + up.N) * q.K, (double)(q.Npad + up.Npad) * q.K * 2);
  const int split = q.Npad / 32;
  const int ftiles = split + up.Npad / 32;
  static const bool mt_on = !(ETD_XENV("ETD_QKV_MT") && atoi(ETD_XENV("ETD_QKV_MT")) == 0);
  // rows from which the 128 x 128 form runs.  Measured (tools/runs3/r3_run21.sh, us per launch, 32 x 32 form / 128 x 128 form): 54 rows 7.2 / 12.3, 128: 9.2 / 12.6,
  // 216: 12.9 / 12.8, 320: 15.8 / 13.3, 432: 18.9 / 13.1, 512: 19.5 / 13.3, 1 728: 40 / 18 -- the two cross at ~220 rows; bit-identical either way
  static const int mt_min = ETD_XENV("ETD_QKV_MT_MIN") ? atoi(ETD_XENV("ETD_QKV_MT_MIN")) : 256;
  if (mt_on && q.M >= mt_min && q.Npad % 128 == 0 && up.Npad % 128 == 0 && q.Wf && up.Wf) {
    const int split128 = q.Npad / 128, ft128 = split128 + up.Npad / 128, RT128 = (q.M + 127) / 128;
    hipLaunchKernelGGL(k_dstep_qkv_up_mt, dim3((unsigned)(((ft128 + 7) / 8) * 8 * RT128)), dim3(256), 0, st, q.M, split128, ft128, (const d16*)q.Wf, (const d16*)up.Wf, q.Xb, up.Xb, q.ldx, up.ldx, q, up);
    HIP_TRY(hipGetLastError());
    return ETD_OK;
  }
  static const int rpt_env = ETD_XENV("ETD_QKV_RPT") ? atoi(ETD_XENV("ETD_QKV_RPT")) : 0;
  const int rpt = rpt_env > 0 ? rpt_env : (q.M > DS_MAX_ROWS ? 4 : 1);
  const int RT = (q.M + 31) / 32, RG = (RT + rpt - 1) / rpt;
  hipLaunchKernelGGL(k_dstep_qkv_up, dim3((unsigned)(((ftiles + 7) / 8) * 8 * RG)), dim3(64 * DS_WAVES), 0, st, q.M, q.K, split, ftiles, q.W, up.W, q.Xb, up.Xb, q.ldx, up.ldx, rpt, q, up);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// Token choice for one row by ONE wave: greedy argmax (lowest index on ties, like torch.argmax), or the reference's
// sampling branch (etude_decoder.py:321-331): p = softmax(logits / T); sort descending; drop every token whose
// predecessors already hold more than top_p of the mass (the first token always stays); renormalise; draw.
// `lg` may be LDS or global; `sp`/`ss`/`si` are per-wave scratch of >= V entries.  V <= 256.
// ================================================================================================
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {      // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ int wave_argmax(const float* lg, int V, int lane) {
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int v = lane; v < V; v += 64) {
    const float x = lg[v];
    if (x > best || (x == best && v < bi)) { best = x; bi = v; }
  }
#define ETD_AMAX_STAGE(O) { const float ob = lane_xor<O>(best); const int oi = lane_xor<O>(bi); \
    const bool take = (ob > best) | ((ob == best) & (oi < bi)); best = take ? ob : best; bi = take ? oi : bi; }
  ETD_AMAX_STAGE(32) ETD_AMAX_STAGE(16) ETD_AMAX_STAGE(8) ETD_AMAX_STAGE(4) ETD_AMAX_STAGE(2) ETD_AMAX_STAGE(1)
#undef ETD_AMAX_STAGE
  return bi;
}
// Four rows at once: the same comparisons as wave_argmax, branch-free and interleaved so that the rows' LDS reads and
// cross-lane exchanges overlap (one row at a time is a chain of ~14 dependent LDS round trips).  V <= 256.
__device__ __forceinline__ void wave_argmax4(const float* lg, int ld, int V, int lane, int (&out)[4]) {
  float best[4]; int bi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = 0x7fffffff; }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = lane + 64 * it;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = lg[j * ld + (v < V ? v : 0)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool take = (v < V) & ((x[j] > best[j]) | ((x[j] == best[j]) & (v < bi[j])));      // bitwise on purpose: && / || became branch ladders
      best[j] = take ? x[j] : best[j]; bi[j] = take ? v : bi[j];
    }
  }
#define ETD_AMAX_STAGE(O) _Pragma("unroll") for (int j = 0; j < 4; ++j) { const float ob = lane_xor<O>(best[j]); const int oi = lane_xor<O>(bi[j]); \
    const bool take = (ob > best[j]) | ((ob == best[j]) & (oi < bi[j])); best[j] = take ? ob : best[j]; bi[j] = take ? oi : bi[j]; }
  ETD_AMAX_STAGE(32) ETD_AMAX_STAGE(16) ETD_AMAX_STAGE(8) ETD_AMAX_STAGE(4) ETD_AMAX_STAGE(2) ETD_AMAX_STAGE(1)
#undef ETD_AMAX_STAGE
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = bi[j];
}
__device__ __attribute__((noinline)) int wave_sample(const float* lg, int V, int lane, float inv_temp, float top_p, unsigned long long seed, unsigned long long key,
                           unsigned ctr, float* sp, float* ss, int* si) {
  // softmax(logits / T) in fp32
  float mx = -INFINITY;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, lg[v] * inv_temp);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int v = lane; v < V; v += 64) { const float e = expf(lg[v] * inv_temp - mx); sp[v] = e; sum += e; }
  sum = wave_sum(sum);
  for (int v = lane; v < V; v += 64) sp[v] = sp[v] / sum;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // rank = position in the descending order (ties: lower index first), scatter into sorted order
  for (int v = lane; v < V; v += 64) {
    const float p = sp[v];
    int rank = 0;
    for (int u = 0; u < V; ++u) { const float q = sp[u]; rank += (q > p || (q == p && u < v)) ? 1 : 0; }
    ss[rank] = p; si[rank] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int tok = 0;
  if (lane == 0) {
    int K = V;
    if (top_p > 0.f && top_p < 1.f) {
      float cum = 0.f;
      K = 1;
      for (int k = 0; k + 1 < V; ++k) {        // token k+1 is removed iff cumsum[k] > top_p (:326-328)
        cum += ss[k];
        if (cum > top_p) break;
        K = k + 2;
      }
    }
    float S = 0.f;
    for (int k = 0; k < K; ++k) S += ss[k];
    const unsigned long long r = mix64(mix64(seed ^ mix64(key)) + (unsigned long long)ctr);
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);                  // 24 random bits -> [0, 1)
    const float target = u * S;
    float acc = 0.f;
    tok = si[K - 1];
    for (int k = 0; k < K; ++k) { acc += ss[k]; if (target < acc) { tok = si[k]; break; } }
  }
  return __shfl(tok, 0, 64);
}

// ================================================================================================
// k_dstep_head: tail of decode step t and head of step t+1 in one launch, one workgroup per 32 rows:
//   final LayerNorm -> lm_head logits (MFMA; eight 64-wide K slices accumulated separately and added in order) -> greedy argmax (lowest index on ties) -> stream
//   state update (etude_decoder.py:333-343) -> embedding of the new token (:166-179), row metadata and the first layer's
//   two LayerNorms for the next step.
// ================================================================================================
// Diagnostic build (-DETD_HEAD_STAMP, tools/head_stamp.py): s_memtime stamps at the phase boundaries of workgroups 0..7,
// read back with etd_debug_head_stamps.  Each stamp costs ~900 clk, so the shipped build has none.
#ifdef ETD_HEAD_STAMP
__device__ long long g_head_stamp[8 * 16];
#define HSTAMP(i) do { if (tid == 0 && blockIdx.x < 8) g_head_stamp[blockIdx.x * 16 + (i)] = clock64(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif
#define DH_LDX 520   // LayerNorm'ed rows in LDS: 512 + 8 d16 (1040 B rows: conflict-free 16-byte fragment reads)
#define DH_LDL 257   // logits rows in LDS (floats): up to 256 vocabulary entries + 1
__global__ __launch_bounds__(512) void k_dstep_head(const float* p_hfin, int p_M, const int* p_row_slot, const int* p_row_active, const int* p_row_pos, DHeadArgs a) {   // leading scalars: kernarg preload
  __shared__ __attribute__((aligned(16))) d16 Xs[32 * DH_LDX];
  __shared__ float Ls[32 * DH_LDL];
  __shared__ float sps[8][256], sss[8][256];
  __shared__ int sis[8][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;   // wave: provably uniform, so the per-row state below is fetched by scalar loads into SGPRs
  const int m0 = blockIdx.x * 32, H = a.H;     // H == 512 (checked by the launcher)
  const float inv_temp = a.samp ? a.samp->inv_temp : 0.f;
  HSTAMP(0);
  // ---- load order (loads return in order; sched barriers pin it): the rows to normalise and the row maps, then the
  // lm_head fragments, LayerNorm, then -- behind the row->slot lookup that has long arrived -- the stream state, whose
  // round trip overlaps the logits phase
  const int k8 = lane * 8;
  // ---- final LayerNorm of 4 rows per wave (summation order of k_dgemm_s's LayerNorm prologue); unrolled: the four rows'
  // loads are in flight together
  f32x4 hv0[4], hv1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int gm = m0 + wave * 4 + j; gm = gm < p_M ? gm : p_M - 1;
    const float* xp = p_hfin + (long long)gm * H;
    hv0[j] = *reinterpret_cast<const f32x4*>(xp + lane * 4); hv1[j] = *reinterpret_cast<const f32x4*>(xp + lane * 4 + 256);
  }
  // row maps of the wave's four rows in ONE load instruction: lane = 4 * field + row (slot, active, position); read back
  // with v_readlane into SGPRs.  (Per-row scalar-style loads were issued and awaited one after the other.)
  int mapv;
  {
    const int j = lane & 3, f = lane >> 2;
    int gm = m0 + wave * 4 + j; gm = gm < p_M ? gm : p_M - 1;
    const int* mp = f == 0 ? p_row_slot : (f == 1 ? p_row_active : p_row_pos);
    mapv = mp[gm];
  }
  f32x4 lg_[2], lb_[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    lg_[half] = *reinterpret_cast<const f32x4*>(a.lnf_g + lane * 4 + half * 256); lb_[half] = *reinterpret_cast<const f32x4*>(a.lnf_b + lane * 4 + half * 256);
  }
  __builtin_amdgcn_sched_barrier(0);
  // the lm_head fragments of this wave's 32 vocabulary entries depend on nothing: all 32 requested here, behind the rows
  // above (loads return in order), and consumed after the LayerNorm -- left to the compiler they were 20 serial round trips
  d16x8 wf[32];
  {
    const bool own = wave * 32 < a.Vpad;                // waves without a tile load one broadcast address (unconditional loads: no merge waits, next to no traffic)
    const d16* wfr = a.Whead + (own ? ((long long)wave * 32 * 64 + lane) * 8 : 0);       // fragment order: [tile][k-step][lane][8], 1 KiB per instruction
#pragma unroll
    for (int s4 = 0; s4 < 32; ++s4) wf[s4] = *reinterpret_cast<const d16x8*>(wfr + s4 * 64 * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    float s4[4], q4[4], mean[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v0 = hv0[j], v1 = hv1[j];
      float s = 0.f;
      s += v0[0] + v0[1] + v0[2] + v0[3];
      s += v1[0] + v1[1] + v1[2] + v1[3];
      s4[j] = s;
    }
    wave_sum_n<4>(s4);
    HSTAMP(6);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v0 = hv0[j], v1 = hv1[j];
      mean[j] = s4[j] / (float)H;
      float q = 0.f;
      { const float d0 = v0[0] - mean[j], d1 = v0[1] - mean[j], d2 = v0[2] - mean[j], d3 = v0[3] - mean[j]; q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3; }
      { const float d0 = v1[0] - mean[j], d1 = v1[1] - mean[j], d2 = v1[2] - mean[j], d3 = v1[3] - mean[j]; q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3; }
      q4[j] = q;
    }
    wave_sum_n<4>(q4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = wave * 4 + j;
      const float rstd = rsqrtf(q4[j] / (float)H + a.eps);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k = lane * 4 + half * 256;
        const f32x4 x = half ? hv1[j] : hv0[j];
        const f32x4 g = lg_[half], b = lb_[half];
        *reinterpret_cast<d16x4*>(Xs + rl * DH_LDX + k) =
            pack4d((x[0] - mean[j]) * rstd * g[0] + b[0], (x[1] - mean[j]) * rstd * g[1] + b[1], (x[2] - mean[j]) * rstd * g[2] + b[2], (x[3] - mean[j]) * rstd * g[3] + b[3]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // stream state of the four rows, again one load instruction: lane = 4 * field + row, fields cur_tok, len, done, n_out,
  // eos, limit; its round trip overlaps the logits phase
  int statev;
  {
    const int j = lane & 3, f = (lane >> 2) < 5 ? (lane >> 2) : 5;
    const int sl = __shfl(mapv, j, 64);
    const int* sp = f == 0 ? a.cur_tok : f == 1 ? a.len : f == 2 ? a.done : f == 3 ? a.n_out : f == 4 ? a.eos : a.limit;
    statev = sp[sl];
  }
  __builtin_amdgcn_sched_barrier(0);
  HSTAMP(7);
  __syncthreads();
  HSTAMP(1);
  // ---- logits: wave w < Vpad/32 owns features [32w, 32w+32)
  if (wave * 32 < a.Vpad) {
    const d16* xrow = Xs + r * DH_LDX;
    f32x16 tot;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = sl * 64 + s4 * 16 + h * 8;
        acc = mfma32(wf[sl * 4 + s4], *reinterpret_cast<const d16x8*>(xrow + k), acc);
      }
      if (sl == 0) tot = acc;
      else {
#pragma unroll
        for (int i = 0; i < 16; ++i) tot[i] += acc[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int f = wave * 32 + acc_row(i, h);
      if (f < a.V) Ls[r * DH_LDL + f] = tot[i];
    }
  }
  __syncthreads();
  HSTAMP(2);
  if (a.logits_dbg) {                      // test hook: this step's logits, row-major [M][V]
    for (int i = tid; i < 32 * a.V; i += 512) {
      const int rr = i / a.V, f = i - rr * a.V;
      if (m0 + rr < a.M) a.logits_dbg[(long long)(m0 + rr) * a.V + f] = Ls[rr * DH_LDL + f];
    }
  }
  // ---- per row: token choice and state update (registers + LDS only), then ONE round trip for the four word-embedding rows
  // ---- token choice, then the state update with ONE LANE PER ROW (lanes 0..3): the fields are gathered from the two
  // load registers by cross-lane reads, and the stores go out per lane.  (Four scalar copies of this ran the kernel out of SGPRs.)
  int gbi[4] = {0, 0, 0, 0};
  if (inv_temp > 0.f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (m0 + wave * 4 + j >= a.M) continue;
      gbi[j] = wave_sample(Ls + (wave * 4 + j) * DH_LDL, a.V, lane, inv_temp, a.samp->top_p, a.samp->seed, a.rng_key[__builtin_amdgcn_readlane(mapv, j)],
                           (unsigned)__builtin_amdgcn_readlane(statev, 12 + j), sps[wave], sss[wave], sis[wave]);
    }
  } else {
    wave_argmax4(Ls + wave * 4 * DH_LDL, DH_LDL, a.V, lane, gbi);
  }
  int n_tok[4], r_slot[4];
  {
    const int jr = lane & 3, m = m0 + wave * 4 + jr;
    const int slot = __shfl(mapv, jr, 64), act = __shfl(mapv, 4 + jr, 64), pos = __shfl(mapv, 8 + jr, 64);
    const int tok = __shfl(statev, jr, 64), len0 = __shfl(statev, 4 + jr, 64), done0 = __shfl(statev, 8 + jr, 64);
    const int n = __shfl(statev, 12 + jr, 64), eos = __shfl(statev, 16 + jr, 64), lim = __shfl(statev, 20 + jr, 64);
    const int bi = jr == 0 ? gbi[0] : jr == 1 ? gbi[1] : jr == 2 ? gbi[2] : gbi[3];
    int ntok = tok, ln = len0, dn = done0;
    const bool mine = (lane < 4) & (m < a.M);
    if (act && !done0) {
      ntok = bi; ln = pos + 1;
      const int fin = ((bi == eos) | (n + 1 >= lim)) ? 1 : 0;
      if (mine) {
        if (n < a.out_cap) a.out_tok[(long long)slot * a.out_cap + n] = bi;
        a.n_out[slot] = n + 1;
        a.cur_tok[slot] = bi;
        a.len[slot] = ln;
        if (fin) a.done[slot] = 1;
      }
      dn = fin;
    }
    // next step's row: position = new length, active = not done
    if (mine) { a.row_pos[m] = ln; a.row_active[m] = dn ? 0 : 1; if (a.row_sp) a.row_sp[2 * m + 1] = ln; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      n_tok[j] = __builtin_amdgcn_readlane(ntok, j); r_slot[j] = __builtin_amdgcn_readlane(mapv, j);
    }
  }
  HSTAMP(3);
  // next embeddings of the four rows: word row + the slot's target projection (k_slot_proj), all in flight together
  f32x4 r_c[2], pg1[2], pb1[2], pg2[2], pb2[2], wv[4][2], pv[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      wv[j][hh] = *reinterpret_cast<const f32x4*>(a.word + (long long)n_tok[j] * H + k8 + 4 * hh);
      pv[j][hh] = *reinterpret_cast<const f32x4*>(a.tgt_proj + (long long)r_slot[j] * H + k8 + 4 * hh);
    }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    r_c[hh] = *reinterpret_cast<const f32x4*>(a.cls_emb + (long long)a.tgt_cls * H + k8 + 4 * hh);
    pg1[hh] = *reinterpret_cast<const f32x4*>(a.g1 + k8 + 4 * hh); pb1[hh] = *reinterpret_cast<const f32x4*>(a.b1 + k8 + 4 * hh);
    pg2[hh] = *reinterpret_cast<const f32x4*>(a.g2 + k8 + 4 * hh); pb2[hh] = *reinterpret_cast<const f32x4*>(a.b2 + k8 + 4 * hh);
  }
  {
    float v[4][8], s4[4], q4[4], mean[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int hh = e >> 2, c = e & 3;
        v[j][e] = (wv[j][hh][c] + r_c[hh][c]) + pv[j][hh][c];          // (word + class) + attribute projection (etude_decoder.py:171-176)
        s1 += v[j][e];
      }
      s4[j] = s1;
    }
    wave_sum_n<4>(s4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mean[j] = s4[j] / (float)H;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d0 = v[j][e] - mean[j]; q += d0 * d0; }
      q4[j] = q;
    }
    wave_sum_n<4>(q4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 4 + j;
      if (m >= a.M) continue;
      const long long ro = (long long)m * H;
      { const f32x4 oa = {v[j][0], v[j][1], v[j][2], v[j][3]}, ob = {v[j][4], v[j][5], v[j][6], v[j][7]};
        *reinterpret_cast<f32x4*>(a.h + ro + k8) = oa; *reinterpret_cast<f32x4*>(a.h + ro + k8 + 4) = ob; }
      const float rstd = rsqrtf(q4[j] / (float)H + a.eps);
      d16x8 o1, o2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o1[e] = (d16)((v[j][e] - mean[j]) * rstd * pg1[e >> 2][e & 3] + pb1[e >> 2][e & 3]);
        o2[e] = (d16)((v[j][e] - mean[j]) * rstd * pg2[e >> 2][e & 3] + pb2[e >> 2][e & 3]);
      }
      *reinterpret_cast<d16x8*>(a.x1 + ro + k8) = o1;
      *reinterpret_cast<d16x8*>(a.x2 + ro + k8) = o2;
    }
  }
  HSTAMP(4);
  SS_FLUSH(0, 0);
}

#ifdef ETD_HEAD_STAMP
extern "C" int etd_debug_head_stamps(long long* out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_head_stamp), sizeof(long long) * 8 * 16));
  return ETD_OK;
}
#endif
int launch_dstep_head(const DHeadArgs& a, hipStream_t st) {
  if (a.samp && !a.rng_key) ETD_FAIL(ETD_EINVAL, "dstep_head: sampling needs stream keys");
  if (a.M < 1 || a.H != 512 || a.V < 1 || a.V > a.Vpad || a.Vpad % 32 || a.Vpad > 256 || !a.hfin || !a.Whead || !a.tgt_proj || !a.h || !a.x1 || !a.x2)
    ETD_FAIL(ETD_EINVAL, "dstep_head: bad arguments (needs hidden 512, vocabulary <= 256)");
  ProfScope ps("k_dstep_head", st, 2.0 * a.M * a.V * a.H, (double)a.Vpad * a.H * 2);
  hipLaunchKernelGGL(k_dstep_head, dim3((a.M + 31) / 32), dim3(512), 0, st, a.hfin, a.M, a.row_slot, a.row_active, a.row_pos, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_dgemv (M == 1, the reference's batch-1 token loop): each wave owns 4 output features, the 64 lanes
// split K in 16-byte pieces (one fully coalesced 1 KiB / 2 KiB row segment per load instruction).
// ================================================================================================
// fmaf as a v_fma_f32 of its own: the SLP vectoriser otherwise pairs the four feature accumulators into v_pk_fma_f32 ... op_sel:[0,1,0]
// (low result reads a high register) -- the packed form kept out of this library, see merge_sum below / tests/test_isa_guard.py
__device__ __forceinline__ float fma_scalar(float a, float b, float c) {
  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
template <bool WBF16, int EPI>
__global__ __launch_bounds__(256) void k_dgemv(DGemmArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (blockIdx.x * 4 + wave) * 4;
  const bool ln = a.ln_g != nullptr;
  float mean = 0.f, rstd = 1.f;
  if (ln) {
    float s = 0.f;
    for (int k = lane * 4; k < a.K; k += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + k); s += v[0] + v[1] + v[2] + v[3]; }
    s = wave_sum(s);
    mean = s / (float)a.K;
    float q = 0.f;
    for (int k = lane * 4; k < a.K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + k);
      const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
      q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    q = wave_sum(q);
    rstd = rsqrtf(q / (float)a.K + a.ln_eps);
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane * 8; k < a.K; k += 512) {
    f32x4 x0 = *reinterpret_cast<const f32x4*>(a.X + k), x1 = *reinterpret_cast<const f32x4*>(a.X + k + 4);
    if (ln) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_g + k), g1 = *reinterpret_cast<const f32x4*>(a.ln_g + k + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + k), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + k + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { x0[j] = (x0[j] - mean) * rstd * g0[j] + b0[j]; x1[j] = (x1[j] - mean) * rstd * g1[j] + b1[j]; }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float w[8];
      if constexpr (WBF16) {
        const d16x8 wv = *reinterpret_cast<const d16x8*>(reinterpret_cast<const d16*>(a.W) + (long long)(nb + f) * a.K + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = bf2f(wv[j]);
        // the MFMA paths round x to d16 as well; keep the GEMV consistent with them
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[f] = fmaf(w[j], bf2f((d16)x0[j]), acc[f]); acc[f] = fmaf(w[4 + j], bf2f((d16)x1[j]), acc[f]); }
      } else {
        const float* wp = reinterpret_cast<const float*>(a.W) + (long long)(nb + f) * a.K + k;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[f] = fma_scalar(w0[j], x0[j], acc[f]); acc[f] = fma_scalar(w1[j], x1[j], acc[f]); }
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = wave_sum(acc[f]);
  if (lane != 0) return;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int n = nb + f;
    if (n >= a.N) break;
    float v = acc[f];
    if (EPI != DEPI_LOGITS) v += a.bias[n];
    if (EPI == DEPI_GELU) v = gelu_erf(v);
    if (EPI == DEPI_RESID) a.hout[n] = (v + a.add[n]) + a.hin[n];
    else a.Y[n] = v;
  }
}

// raw fused-QKV row(s) [M][3H] -> RoPE'd Q [M][H], K/V cache rows (used after the GEMV path)
template <typename KVT>
__global__ void k_rope_scatter(DGemmArgs a, const float* __restrict__ raw) {
  const int m = blockIdx.x;
  const int pos = a.rows.pos[m], slot = a.rows.slot[m];
  const bool act = a.rows.active[m] && pos < a.max_ctx;
  const int H3 = a.n_heads * 192;
  for (int n = threadIdx.x; n < H3; n += blockDim.x) {
    const int head = n / 192, j = n - head * 192, part = j >> 6, d = j & 63;
    float v = raw[(long long)m * H3 + n];
    if (part < 2 && d < 2 * a.rot_half) {
      const int dd = d < a.rot_half ? d : d - a.rot_half;
      const float c = a.rope_cos[(long long)pos * a.rot_half + dd], s = a.rope_sin[(long long)pos * a.rot_half + dd];
      const float other = raw[(long long)m * H3 + (d < a.rot_half ? n + a.rot_half : n - a.rot_half)];
      v = d < a.rot_half ? v * c - other * s : v * c + other * s;
    }
    if (part == 0) a.Q[(long long)m * (a.n_heads * 64) + head * 64 + d] = v;
    else if (act) {
      const long long off = (long long)slot * a.slot_stride + ((long long)head * a.max_ctx + pos) * 64 + d;
      KVT* base = reinterpret_cast<KVT*>(part == 1 ? a.Kc : a.Vc);
      base[off] = (KVT)v;
    }
  }
}

template <bool WBF16>
static void dgemm_dispatch(const DGemmArgs& a, int epi, int path, hipStream_t st) {
  // path 0: 128-feature tile (prefill), 1: skinny K-split tile (M <= DS_MAX_ROWS), 2: GEMV (M == 1)
  const int wtiles = (a.Npad / 32) * (a.k_splits > 1 ? a.k_splits : 1);
  const dim3 g0((a.M + 31) / 32, a.Npad / 128), g1((unsigned)(((wtiles + 7) / 8) * 8 * ((a.M + 31) / 32))), g2(a.Npad / 16);
#define ETD_DG(E)                                                                                   \
  if (path == 0) hipLaunchKernelGGL((k_dgemm<WBF16, E>), g0, dim3(256), 0, st, a);                  \
  else if (path == 1) hipLaunchKernelGGL((k_dgemm_s<WBF16, E>), g1, dim3(64 * DS_WAVES), 0, st, a.M, a.Npad, a.k_splits, a.K, a.W, a.Xb, a.ldx, a);  \
  else hipLaunchKernelGGL((k_dgemv<WBF16, E>), g2, dim3(256), 0, st, a);
  switch (epi) {
    case DEPI_BIAS: ETD_DG(DEPI_BIAS) break;
    case DEPI_GELU: ETD_DG(DEPI_GELU) break;
    case DEPI_RESID: ETD_DG(DEPI_RESID) break;
    case DEPI_LOGITS: ETD_DG(DEPI_LOGITS) break;
    case DEPI_PARTIAL: hipLaunchKernelGGL((k_dgemm_s<WBF16, DEPI_PARTIAL>), g1, dim3(64 * DS_WAVES), 0, st, a.M, a.Npad, a.k_splits, a.K, a.W, a.Xb, a.ldx, a); break;
    default:
      if (path == 0) hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_QKV>), g0, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_dgemm_s<WBF16, DEPI_QKV>), g1, dim3(64 * DS_WAVES), 0, st, a.M, a.Npad, a.k_splits, a.K, a.W, a.Xb, a.ldx, a);
      break;
  }
#undef ETD_DG
}

bool g3_enabled();                                                   // csrc/gemm3.hip: ETD_NO_GEMM3, read once for every call site
bool gemm3_s_takes(const DGemmArgs& a, int epi);                     // csrc/gemm3.hip: the fp32 mode's small-M GEMM on the f16 matrix cores
int launch_gemm3_s(const DGemmArgs& a, int epi, hipStream_t st);
int launch_dgemm(const DGemmArgs& a, int epi, bool w_bf16, hipStream_t st) {
  const bool g3s_on = g3_enabled();
  if (!w_bf16 && g3s_on && gemm3_s_takes(a, epi)) return launch_gemm3_s(a, epi, st);
  if (a.M <= 0 || a.Npad % 128 || a.K % 256 || a.N > a.Npad) ETD_FAIL(ETD_EINVAL, "dgemm: bad shape M=%d N=%d Npad=%d K=%d", a.M, a.N, a.Npad, a.K);
  if (epi == DEPI_QKV && (a.rot_half != 8 || a.N % 192)) ETD_FAIL(ETD_EINVAL, "dgemm: QKV epilogue needs head_dim 64 and rotary_ndims 16");
  if (epi == DEPI_RESID && (a.N % 4)) ETD_FAIL(ETD_EINVAL, "dgemm: resid needs N %% 4 == 0");
  // (a split-K request is a decode-step / last-rows GEMM: skinny tile up to DS_STEP_MAX_ROWS rows)
  const int path = a.M == 1 ? 2 : (a.M <= DS_MAX_ROWS || (epi == DEPI_PARTIAL && a.M <= DS_STEP_MAX_ROWS) ? 1 : 0);
  if (epi == DEPI_PARTIAL && (path != 1 || a.k_splits < 1 || (a.K / a.k_splits) % (64 * DS_WAVES) || !a.Y)) ETD_FAIL(ETD_EINVAL, "dgemm: bad split-K request");
  if (epi != DEPI_PARTIAL && a.k_splits > 1) ETD_FAIL(ETD_EINVAL, "dgemm: k_splits needs DEPI_PARTIAL");
  ProfScope ps(path == 2 ? "k_dgemv" : (path == 1 ? "k_dgemm_s" : "k_dgemm"), st, 2.0 * a.M * a.N * a.K, (double)a.Npad * a.K * (w_bf16 ? 2 : 4));
  if (path == 2 && epi == DEPI_QKV) {
    // GEMV writes the raw fused row into a.Y (caller-provided scratch [3H]); RoPE + Q/K/V scatter follow
    if (!a.Y) ETD_FAIL(ETD_EINVAL, "dgemm: M == 1 QKV path needs a scratch row in Y");
    if (w_bf16) { dgemm_dispatch<true>(a, DEPI_BIAS, 2, st); hipLaunchKernelGGL(k_rope_scatter<d16>, dim3(1), dim3(256), 0, st, a, a.Y); }
    else { dgemm_dispatch<false>(a, DEPI_BIAS, 2, st); hipLaunchKernelGGL(k_rope_scatter<float>, dim3(1), dim3(256), 0, st, a, a.Y); }
  } else if (w_bf16) dgemm_dispatch<true>(a, epi, path, st);
  else dgemm_dispatch<false>(a, epi, path, st);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_dattn: causal attention of each row's query against its slot's KV cache [0, pos]
//                                    modeling_gpt_neox.py:172-190 (softmax in fp32), :222-236
// grid (M, heads); 4 waves split the key blocks; a wave-iteration covers 8 keys: lane = (key j = lane>>3,
// 8-dim chunk c = lane&7), so K/V loads are fully coalesced 2 KB (fp32) / 1 KB (d16) blocks.
// ================================================================================================
// raw 8-element K/V pieces stay in their memory format in registers (so the loads remain in flight) and are widened at use
template <typename KVT> struct Raw8;
template <> struct Raw8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
};
template <> struct Raw8<d16> {
  d16x8 v;
  // K/V rows are read once per step and never again before they are overwritten in the caches by the next row's stream: the
  // nontemporal hint keeps them from evicting the weights (shared by all engines) from L2 / the Infinity Cache -- measured
  // -5.5 % per step with one engine and with four (tools/bench_engine_overlap.py)
  __device__ __forceinline__ void load(const d16* p) { v = __builtin_nontemporal_load(reinterpret_cast<const d16x8*>(p)); }
  __device__ __forceinline__ float get(int j) const { return bf2f(v[j]); }
};

// ================================================================================================
// Row finish (DRowFin): what k_resid_ln_rows<12> does for ONE row, by ONE wave, inside the launch that produced the slabs.
// The slabs were stored write-through (sc1) by other workgroups, possibly on other XCDs, and every one of them added to the
// row's counter after its stores had drained; this wave's add came last, so every load of them here is an sc1 load (past
// this CU's L1, which other CUs' stores never refresh) -- MI355X_MICROARCH.md, inter-workgroup visibility, "all sc1" form.
// The sums run in slab order like the row kernel's, so the result is bit-identical to it.
// ================================================================================================
#ifndef ETD_FIN_ZB
#define ETD_FIN_ZB 12      // slabs requested per batch by the finishing wave: all of them (96 registers in flight)
#endif
typedef __amdgpu_buffer_rsrc_t drsrc_t;
__device__ __forceinline__ drsrc_t d_rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000);
}
#define D_SC1 16     // cache-policy bit of the raw buffer intrinsics on gfx942 / gfx950: sc1
__device__ __forceinline__ void row_finish(const DRowFin& f, const int row_in, const int M_in, const int lane) {
  // the row is the same for the whole wave: say so, or hipcc wraps every buffer load in a readfirstlane loop with its own vmcnt(0)
  // (24 dependent round trips: the first build of this function took 22 us per row)
  const int row = __builtin_amdgcn_readfirstlane(row_in), M = __builtin_amdgcn_readfirstlane(M_in);
  const drsrc_t ps = d_rsrc(f.P, (long long)f.nslab * M * 512 * 4);
  const long long ro = (long long)row * 512;
  const int k = lane * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // (parameters and the residual row come from earlier launches: plain loads, requested together with the first batch)
  const f32x4 ba = *reinterpret_cast<const f32x4*>(f.bias + k), bb = *reinterpret_cast<const f32x4*>(f.bias + k + 4);
  const f32x4 ha = *reinterpret_cast<const f32x4*>(f.hin + ro + k), hb = *reinterpret_cast<const f32x4*>(f.hin + ro + k + 4);
  f32x4 pg1[2], pb1[2], pg2[2], pb2[2];
  if (f.x1) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      pg1[hh] = *reinterpret_cast<const f32x4*>(f.g1 + k + 4 * hh); pb1[hh] = *reinterpret_cast<const f32x4*>(f.b1 + k + 4 * hh);
      pg2[hh] = *reinterpret_cast<const f32x4*>(f.g2 + k + 4 * hh); pb2[hh] = *reinterpret_cast<const f32x4*>(f.b2 + k + 4 * hh);
    }
  }
  constexpr int ZB = ETD_FIN_ZB;   // slabs per batch of loads
#pragma unroll
  for (int z0 = 0; z0 < 12; z0 += ZB) {
    f32x4 pa[ZB], pb[ZB];
#pragma unroll
    for (int z = 0; z < ZB; ++z) {
      const int so = (int)((((long long)(z0 + z) * M + row) * 512) * 4);       // < 2^31: 12 slabs x 512 rows x 2 KiB
      pa[z] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ps, k * 4, so, D_SC1));
      pb[z] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ps, k * 4 + 16, so, D_SC1));
    }
#pragma unroll
    for (int z = 0; z < ZB; ++z)
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] += pa[z][j]; acc[4 + j] += pb[z][j]; }
  }
  float v[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = ((acc[j] + ba[j]) + 0.f) + ha[j]; v[4 + j] = ((acc[4 + j] + bb[j]) + 0.f) + hb[j]; }   // (+ 0.f: the row kernel's absent `add` term)
  {
    const f32x4 oa = {v[0], v[1], v[2], v[3]}, ob = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(f.hout + ro + k) = oa;
    *reinterpret_cast<f32x4*>(f.hout + ro + k + 4) = ob;
  }
  if (!f.x1) return;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  s = wave_sum(s);
  const float mean = s / 512.f;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float d0 = v[j] - mean; q += d0 * d0; }
  q = wave_sum(q);
  const float rstd = rsqrtf(q / 512.f + f.eps);
  d16x8 o1, o2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float n = (v[j] - mean) * rstd;
    o1[j] = (d16)(n * pg1[j >> 2][j & 3] + pb1[j >> 2][j & 3]);
    o2[j] = (d16)(n * pg2[j >> 2][j & 3] + pb2[j >> 2][j & 3]);
  }
  *reinterpret_cast<d16x8*>(f.x1 + ro + k) = o1;
  *reinterpret_cast<d16x8*>(f.x2 + ro + k) = o2;
}
#define D_ARRIVE(p) __hip_atomic_fetch_add((p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define D_CNT_RESET(p) __hip_atomic_store((p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// measurement builds ((history: 4ac2f57) tools/runs/r2_run18.sh): -DETD_ABL_DENSE / -DETD_ABL_GEMMW / -DETD_ABL_QKVW replace a weight stream by a
// constant (wrong results on purpose) to see what that stream costs the OTHER engines' kernels
#define ABL_DENSE_LOAD(p) (*reinterpret_cast<const d16x8*>(p))
#define ABL_GEMMW_LOAD(p) (*reinterpret_cast<const d16x8*>(p))
// cross-lane exchanges of the attention core: 0 = __shfl_xor (ds_bpermute_b32, LDS crossbar), 1 = lane_xor (DPP / permlane swaps)
#define ETD_XCH(v, O) __shfl_xor(v, O, 64)
// measurement builds: ETD_AD_TRANS_NOP=1 puts 16 wait states between the merge's v_exp_f32 results and their first use, = 2 also in the key loop
#define ETD_TRANS_SETTLE(a_, b_) ((void)0)
#define ETD_TRANS_SETTLE_LOOP(a_, b_) ((void)0)
#define EXPF(x) (FAST ? __builtin_amdgcn_exp2f(x) : expf(x))
// DENSE: instead of storing the head's 64 outputs, multiply them (rounded to d16, as the projection GEMM would read them)
// with this head's [512][64] slice of attention.dense and store the 512 partial sums as one more split-K slab for
// k_resid_ln_rows -- the attention-output projection needs no launch of its own (k_dstep_attn_down below).
//
// NW = waves per (row, head) workgroup.  A wave-iteration covers 16 keys and keeps the next iteration's K/V requested, so a
// workgroup has 2 x 16 NW keys x 256 B in flight and walks the context in ceil(ctx / (16 NW)) dependent round trips: at the
// serving shape (54 rows per engine, ctx ~340) NW = 4 is six round trips of ~1-2 us each with ~30 KiB in flight per CU (the
// 0.29-of-peak kernel of round 1); NW = 16 requests the whole context of a (row, head) at once (2 iterations, both in flight).
// a * b + c * d of the softmax merge as three VALU instructions of its own (mul, mul, add: the roundings hipcc's code has under
// -ffp-contract=off).  Left to the SLP vectoriser these sums are paired crosswise -- v_pk_mul_f32 x2 + v_pk_add_f32 / v_pk_fma_f32
// with op_sel:[0,0,1] op_sel_hi:[1,1,0]: the LOW result takes an operand's HIGH register and vice versa -- and on MI355X such an
// instruction's low result came out as if that operand were 0 in lanes 48-55 whenever the SIMD was shared with another queue's
// MFMA waves (batched prefill, Extract stage): a (wave, j = 6) slot's softmax denominator vanished, the head's output grew by
// ~3 %, tokens changed from run to run.  LABNOTES.md, "packed FP32 with crossed op_sel"; tools/probe_trace.py;
// tests/test_isa_guard.py keeps the form out of the library.
// (-DETD_AD_CROSSED_PK=1 rebuilds the failing form: tools/probe_trace.py and tests/test_gpu_reproducibility.py then fail again)
__device__ __forceinline__ float merge_sum(float a, float b, float c, float d) {
  float t, u, r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u) : "v"(c), "v"(d));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(u));
  return r;
}
// PAIR (DENSE only; the default of the decode step, ETD_AD_PAIR=0 turns it off): the workgroup holds TWO rows of the same head, NW waves each (threads 0 .. 64 NW - 1
// row `m`, the rest row `m + 1`, given by the caller with its own `red` / `osh` / `outsh`); every one of the 2 NW waves then applies 512 / (2 NW)
// rows of the head's dense slice to BOTH rows' outputs, so the slice is read once per pair (half the L2 traffic of the dense phase, one
// pass of 8 fragments instead of two) -- per row the same operations in the same order: bit-identical.  `dup`: the second row repeats
// the first (odd M) and stores nothing.
template <typename KVT, bool DENSE, int NW, bool FIN = false, bool PAIR = false>
__device__ __forceinline__ void dattn_core(const int m, const int head, float (&red)[NW][8][10], float* osh, float* outsh,
                                           const int* p_row_sp, const float* p_Q, const void* p_Kc, const void* p_Vc, long long p_slot_stride,
                                           int p_max_ctx, int p_n_heads, float p_scale, int p_identity, const DAttnArgs& a, const DRowFin* fin = nullptr,
                                           const float* osh_other = nullptr, float* outsh_other = nullptr, bool dup = false) {
  const int tid = PAIR ? (int)(threadIdx.x & (64 * NW - 1)) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 3, c = lane & 7;
  SS_DECL(); SS(0);
  constexpr int G2 = 8 * NW;         // offset of a wave's second 8-key group inside an iteration
  constexpr int KI = 16 * NW;        // keys per workgroup iteration
  int slot, pos;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  // p_identity (row i is slot i, known on the host): the K/V addresses of the first key block then depend on nothing in
  // memory, so that block is requested BEFORE the position arrives -- the row-metadata round trip and the first K/V round
  // trip overlap instead of following each other (keys past the context are fetched from the slot's own cache and ignored).
  Raw8<KVT> kA, kB, wA, wB, nkA, nkB, nwA, nwB;
  if (p_identity) {
    const KVT* kb0 = reinterpret_cast<const KVT*>(p_Kc) + (long long)m * p_slot_stride + (long long)head * p_max_ctx * 64;
    const KVT* vb0 = reinterpret_cast<const KVT*>(p_Vc) + (long long)m * p_slot_stride + (long long)head * p_max_ctx * 64;
    const int ka = wave * 8 + j, kbb = ka + G2;                        // < 16 NW <= max_ctx (checked by the launcher)
    kA.load(kb0 + (long long)ka * 64 + c * 8); kB.load(kb0 + (long long)kbb * 64 + c * 8);
    wA.load(vb0 + (long long)ka * 64 + c * 8); wB.load(vb0 + (long long)kbb * 64 + c * 8);
    slot = m; pos = __builtin_nontemporal_load(p_row_sp + 2 * m + 1);
  } else if (p_row_sp) { const i32x2 sp = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(p_row_sp + 2 * m)); slot = sp[0]; pos = sp[1]; }   // one 8-byte load
  else { slot = a.rows.slot[m]; pos = a.rows.pos[m]; }
  const int ctx = (pos < p_max_ctx ? pos : p_max_ctx - 1) + 1;
  const int hidden = p_n_heads * 64;
  // d16 serving mode: scores pre-scaled by log2(e) and v_exp_f32 (exp2) -- softmax is base-invariant; the fp32
  // parity mode keeps the accurate expf like torch's softmax
  constexpr bool FAST = sizeof(KVT) == 2;
  const float qs = FAST ? p_scale * 1.4426950408889634f : p_scale;
  float q[8];
  {
    const float* qp = p_Q + (long long)m * hidden + head * 64 + c * 8;
    const f32x4 x = *reinterpret_cast<const f32x4*>(qp), y = *reinterpret_cast<const f32x4*>(qp + 4);
    q[0] = x[0] * qs; q[1] = x[1] * qs; q[2] = x[2] * qs; q[3] = x[3] * qs;
    q[4] = y[0] * qs; q[5] = y[1] * qs; q[6] = y[2] * qs; q[7] = y[3] * qs;
  }
  const KVT* kb = reinterpret_cast<const KVT*>(p_Kc) + (long long)slot * p_slot_stride + (long long)head * p_max_ctx * 64;
  const KVT* vb = reinterpret_cast<const KVT*>(p_Vc) + (long long)slot * p_slot_stride + (long long)head * p_max_ctx * 64;
  float mr = -INFINITY, lr = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  // Software-pipelined KV stream: a wave-iteration covers two 8-key groups (16 keys); the K/V pieces of iteration i+1
  // are requested before iteration i is consumed, so ~8 KB per wave are always in flight.
  auto issue = [&](int k0, Raw8<KVT>& a_, Raw8<KVT>& b_, Raw8<KVT>& c_, Raw8<KVT>& d_) {
    int ka = k0 + j, kbb = k0 + G2 + j;
    ka = ka < ctx ? ka : ctx - 1; kbb = kbb < ctx ? kbb : ctx - 1;
    a_.load(kb + (long long)ka * 64 + c * 8);
    b_.load(kb + (long long)kbb * 64 + c * 8);
    c_.load(vb + (long long)ka * 64 + c * 8);
    d_.load(vb + (long long)kbb * 64 + c * 8);
  };
  int k0 = wave * 8;
  if (!p_identity && k0 < ctx) issue(k0, kA, kB, wA, wB);
#ifdef ETD_STEP_STAMP
  SS_LANDED(); SS(1);             // row metadata, q and the first K/V block have arrived
#endif
  for (; k0 < ctx; k0 += KI) {
    const bool more = k0 + KI < ctx;
    if (more) issue(k0 + KI, nkA, nkB, nwA, nwB);
    const bool vA = k0 + j < ctx, vB = k0 + G2 + j < ctx;
    float sA = 0.f, sB = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sA = fmaf(q[e], kA.get(e), sA); sB = fmaf(q[e], kB.get(e), sB); }
    sA += ETD_XCH(sA, 1); sB += ETD_XCH(sB, 1);      // ds_bpermute on purpose: the LDS pipe is idle here and the VALU is not (DPP measured 6 % slower)
    sA += ETD_XCH(sA, 2); sB += ETD_XCH(sB, 2);
    sA += ETD_XCH(sA, 4); sB += ETD_XCH(sB, 4);
    if (vA) {
      const float mn = fmaxf(mr, sA);
      float al = EXPF(mr - mn), p = EXPF(sA - mn);
      ETD_TRANS_SETTLE_LOOP(al, p);
      lr = lr * al + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * al + p * wA.get(e);
      mr = mn;
    }
    if (vB) {
      const float mn = fmaxf(mr, sB);
      float al = EXPF(mr - mn), p = EXPF(sB - mn);
      ETD_TRANS_SETTLE_LOOP(al, p);
      lr = lr * al + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * al + p * wB.get(e);
      mr = mn;
    }
    if (more) { kA = nkA; kB = nkB; wA = nwA; wB = nwB; }
  }

  // DENSE: this wave's dense-weight fragments (512 / NW rows = 64 / NW one-KiB fragments; the first 8 of them at most) are
  // requested here -- the key loop's K/V registers are dead, and the merge below (shuffles, LDS, a barrier) covers their round trip
  SS(2);                           // key loop done
  constexpr int FR = PAIR ? 64 / (2 * NW) : 64 / NW, FP = FR > 8 ? 8 : FR, NPASS = FR / FP;
  const int dwave = PAIR ? (int)(threadIdx.x >> 6) : wave;             // wave index among the waves that share the dense slice
  constexpr int DROWS = PAIR ? 512 / (2 * NW) : 512 / NW;              // dense rows (output features) per wave
  d16x8 dwv[DENSE ? FP : 1];
  const d16* dwb = nullptr;
  // measurement builds: ETD_AD_DENSE_LATE=1 requests the fragments AFTER the intra-wave merge, =2 requests them here but drains them before the merge
  if constexpr (DENSE) {
    dwb = a.dense_w + (long long)head * (512 * 64) + (long long)(dwave * DROWS + (lane >> 3)) * 64 + (lane & 7) * 8;
#pragma unroll
    for (int it = 0; it < FP; ++it) dwv[it] = ABL_DENSE_LOAD(dwb + it * 8 * 64);
  }
  // merge the 8 key slots of this wave (lanes differing in bits 3..5), then the NW waves through LDS
#define ETD_MERGE_STAGE(off)                                                                                                \
  {                                                                                                                         \
    const float m2 = ETD_XCH(mr, off), l2 = ETD_XCH(lr, off);                                                               \
    const float mn = fmaxf(mr, m2);                                                                                         \
    float f1 = (mr == -INFINITY) ? 0.f : EXPF(mr - mn), f2 = (m2 == -INFINITY) ? 0.f : EXPF(m2 - mn);                       \
    ETD_TRANS_SETTLE(f1, f2);                                                                                               \
    lr = merge_sum(lr, f1, l2, f2);                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) { const float o2 = ETD_XCH(o[e], off); o[e] = merge_sum(o[e], f1, o2, f2); } \
    mr = mn;                                                                                                                \
  }
#define ETD_MERGE_STAGES() ETD_MERGE_STAGE(8) ETD_MERGE_STAGE(16) ETD_MERGE_STAGE(32)
  float* dp = a.dbg ? a.dbg + ((long long)(head * a.M + m) * 256 + tid) * 8 : nullptr;
  if (dp) { dp[0] = lr; dp[1] = mr; dp[2] = o[0]; }
  ETD_MERGE_STAGES()
  if (dp) { dp[5] = lr; dp[7] = o[0]; }
  if (j == 0) {
    red[wave][c][0] = mr; red[wave][c][1] = lr;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c][2 + e] = o[e];
  }
  __syncthreads();
  if (wave == 0) {
    if constexpr (NW <= 4) {
      // (the round-1 order of operations, kept bit for bit: NW = 4 is what the goldens of the d16 mode were taken with)
      if (j == 0) {
        float M2 = -INFINITY;
        for (int w = 0; w < NW; ++w) M2 = fmaxf(M2, red[w][c][0]);
        float L = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < NW; ++w) {
          const float mw = red[w][c][0];
          const float f = (mw == -INFINITY) ? 0.f : EXPF(mw - M2);
          L += red[w][c][1] * f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += red[w][c][2 + e] * f;
        }
        lr = L;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[e];
      }
    } else {
      // lane (j, c) folds waves j, j + 8 (fixed order), then the same three exchange stages as above fold the 8 partial results
      mr = red[j][c][0]; lr = red[j][c][1];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = red[j][c][2 + e];
#pragma unroll
      for (int w = j + 8; w < NW; w += 8) {
        const float m2 = red[w][c][0], l2 = red[w][c][1];
        const float mn = fmaxf(mr, m2);
        const float f1 = (mr == -INFINITY) ? 0.f : EXPF(mr - mn), f2 = (m2 == -INFINITY) ? 0.f : EXPF(m2 - mn);
        lr = merge_sum(lr, f1, l2, f2);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = merge_sum(o[e], f1, red[w][c][2 + e], f2);
        mr = mn;
      }
      ETD_MERGE_STAGES()
    }
    if (j == 0) {
      const float inv = 1.f / lr;
      const f32x4 x = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv}, y = {o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv};
      if constexpr (DENSE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { osh[c * 8 + e] = (float)(d16)x[e]; osh[c * 8 + 4 + e] = (float)(d16)y[e]; }
      } else {
        float* op = a.O + (long long)m * hidden + head * 64 + c * 8;
        *reinterpret_cast<f32x4*>(op) = x;
        *reinterpret_cast<f32x4*>(op + 4) = y;
        if (a.Ob) {
          const d16x8 ob = {(d16)x[0], (d16)x[1], (d16)x[2], (d16)x[3], (d16)y[0], (d16)y[1], (d16)y[2], (d16)y[3]};
          *reinterpret_cast<d16x8*>(a.Ob + (long long)m * (a.ldob > 0 ? a.ldob : hidden) + head * 64 + c * 8) = ob;
        }
      }
    }
  }
#undef ETD_MERGE_STAGES
#undef ETD_MERGE_STAGE
  if constexpr (DENSE) {
    // out[n] = sum_d Wd[n][64 head + d] * o[d] for the 512 outputs: 8 lanes share a weight row (128 contiguous bytes), a wave
    // instruction covers 8 consecutive rows = 1 KiB; the wave's fragments are in flight together (requested above, before
    // the merge), the 8-lane sums use DPP exchanges.  The weights of a head are one contiguous 64 KiB block (dense_w).
    // (register budget at NW = 4: the kernel streams K/V at 7 waves per SIMD, so its 16 fragments come in two passes of 8)
    __syncthreads();
    SS(3);                         // merged across waves, normalised
    const int g8 = lane >> 3, sub = lane & 7;
    float ov[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = osh[sub * 8 + e];
    if constexpr (PAIR) {
      // this wave's 8 fragments (64 output features) against this half's row AND the other half's
      float ov2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ov2[e] = osh_other[sub * 8 + e];
#pragma unroll
      for (int it = 0; it < FP; ++it) {
        float sacc = 0.f, sacc2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sacc = fmaf(bf2f(dwv[it][e]), ov[e], sacc); sacc2 = fmaf(bf2f(dwv[it][e]), ov2[e], sacc2); }
        sacc += lane_xor<1>(sacc); sacc += lane_xor<2>(sacc); sacc += lane_xor<4>(sacc);
        sacc2 += lane_xor<1>(sacc2); sacc2 += lane_xor<2>(sacc2); sacc2 += lane_xor<4>(sacc2);
        if (sub == 0) { outsh[dwave * DROWS + it * 8 + g8] = sacc; outsh_other[dwave * DROWS + it * 8 + g8] = sacc2; }
      }
    } else
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      if (pass == 1) {
#pragma unroll
        for (int it = 0; it < FP; ++it) dwv[it] = ABL_DENSE_LOAD(dwb + (FP + it) * 8 * 64);
      }
#pragma unroll
      for (int it = 0; it < FP; ++it) {
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc = fmaf(bf2f(dwv[it][e]), ov[e], sacc);
        sacc += lane_xor<1>(sacc); sacc += lane_xor<2>(sacc); sacc += lane_xor<4>(sacc);
        if (sub == 0) outsh[wave * (512 / NW) + (pass * FP + it) * 8 + g8] = sacc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    SS(4);                         // dense slice applied
    if constexpr (FIN) {
      // the head's slab row leaves write-through in 16-byte pieces; every storing wave drains, then ONE lane arrives at the row's counter
      if (tid < 128) {
        const drsrc_t os = d_rsrc(a.dense_out, (long long)p_n_heads * a.M * 512 * 4);
        const f32x4 o4 = {outsh[4 * tid], outsh[4 * tid + 1], outsh[4 * tid + 2], outsh[4 * tid + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), os, tid * 16, (int)((((long long)head * a.M + m) * 512) * 4), D_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      SS(5); SS_FLUSH(2, 0);
      if (wave != 0) return;
#ifdef ETD_STEP_STAMP
      ss_t[0] = ss_t[5]; ss_t[3] = ss_t[4] = ss_t[5] = 0;
#endif
      int old = 0;
      if (lane == 0) old = D_ARRIVE(fin->cnt + m);
      old = __builtin_amdgcn_readfirstlane(old);
      SS(1);
#ifdef ETD_STEP_STAMP
      if (old != fin->target - 1) { SS_FLUSH(4, 0); return; }
#else
      if (old != fin->target - 1) return;
#endif
      row_finish(*fin, m, a.M, lane);
      if (lane == 0) D_CNT_RESET(fin->cnt + m);
#ifdef ETD_STEP_STAMP
      SS_LANDED(); SS(2); SS_FLUSH(4, 1);
#endif
    } else {
      if (tid < 256 && !dup) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 o2 = {outsh[2 * tid], outsh[2 * tid + 1]};
        *reinterpret_cast<f32x2*>(a.dense_out + ((long long)head * a.M + m) * 512 + 2 * tid) = o2;
      }
      SS(5); SS_FLUSH(2, 0);
    }
  }
}
// device-side span stamps of a launch (`a.stamp`, null in every production graph; layout and protocol: k_dstep_attn_down below)
__device__ __forceinline__ void attn_stamp_begin(unsigned long long* stp, int par, int flat_block) {
  if (stp && flat_block == 0 && threadIdx.x < 64) {
    const int bank = par & 1, prev = bank ^ 1, lane = threadIdx.x;
    const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    unsigned long long e = atomicExch(stp + 8 + 64 * prev + lane, 0ull);          // read-and-clear at L2, 64 slots in one round trip
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(e, o, 64); e = x > e ? x : e; }
    if (lane == 0) {
      const unsigned long long s0 = atomicExch(stp + 2 + prev, 0ull);
      if (s0 != 0 && e > s0) {
        atomicAdd(stp, e - s0);
        const unsigned long long n = atomicAdd(stp + 1, 1ull);
        if (n < ETD_STAMP_LOGCAP) { stp[ETD_STAMP_HDR + 2 * n] = s0; stp[ETD_STAMP_HDR + 2 * n + 1] = e; }      // the launch's own (start, end), for unions across engines
      }
      atomicExch(stp + 2 + bank, now);
    }
  }
}
__device__ __forceinline__ void attn_stamp_end(unsigned long long* stp, int par, int flat_block) {
  if (stp) {
    __syncthreads();                     // every wave of the workgroup is back from the body (its early exits return here)
    if (threadIdx.x == 0) atomicMax(stp + 8 + 64 * (par & 1) + (flat_block & 63), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

template <typename KVT>
// (leading scalar arguments: gfx950 preloads the first 16 kernarg dwords into SGPRs at wave launch, so the prologue's address
// arithmetic starts without the kernarg s_load round trip; the struct carries everything that is needed later)
__global__ __launch_bounds__(256) void k_dattn(const int* p_row_sp, const float* p_Q, const void* p_Kc, const void* p_Vc, long long p_slot_stride,
                                               int p_max_ctx, int p_n_heads, float p_scale, int p_identity, DAttnArgs a) {
  __shared__ float red[4][8][10];            // per wave, per dim-chunk: m, l, o[8]
  const int flat = blockIdx.y * gridDim.x + blockIdx.x;
  attn_stamp_begin(a.stamp, a.stamp_par, flat);
  dattn_core<KVT, false, 4>(blockIdx.x, blockIdx.y, red, nullptr, nullptr, p_row_sp, p_Q, p_Kc, p_Vc, p_slot_stride, p_max_ctx, p_n_heads, p_scale, p_identity, a);
  attn_stamp_end(a.stamp, a.stamp_par, flat);
}

#undef EXPF
// ================================================================================================
// k_dstep_attn_down<NW>: one launch for the two things that depend on the QKV|up launch only
//   * workgroups [0, p_gemm_wgs): the MLP down projection as split-K partial slabs.  A workgroup is NW / 2 independent
//     2-wave units of k_dgemm_s's d16 path (32x32 tile, one K slab of 512, K halved over the unit's two waves); the
//     units of a workgroup are consecutive slots of one XCD, i.e. row tiles of the same weight tile where there are several.
//   * the others: attention per (row, head), NW waves, with the head's slice of attention.dense applied in place (dattn_core<DENSE>).
// k_resid_ln_rows then sums k_splits + n_heads slabs.  A decode-step layer is 3 launches instead of 4.
// ================================================================================================
template <int NW> struct AdOcc { static constexpr int lo = 7, hi = 8; };      // 72 registers; forcing 64 (8 waves per SIMD) spills inside the key loop, and a scratch reload there drains the K/V stream
#ifndef ETD_AD_OCC
#define ETD_AD_OCC 5      // built for 5 waves per SIMD (96 registers).  Job level (bench.py, (history: 4ac2f57) tools/runs/r2_run28.sh): 7 -> 581-583, 5 -> 586-588, 4 -> 588, 3 -> 585 audio-s/s
#endif
template <> struct AdOcc<4> { static constexpr int lo = ETD_AD_OCC, hi = 8; };
#ifndef ETD_FIN_OCC
#define ETD_FIN_OCC 4      // waves per SIMD the row-finish instantiation is built for (128 registers).  Job-level (bench.py, r2_run27.sh): 4 -> 567, 5 -> 552, 6 -> 515-522 audio-s/s; at 7 (72 registers) the key loop spills
#endif
template <int NW, bool FIN, bool PAIR = false>      // PAIR (NW = 8 threads-wise): attention workgroups hold two rows of a head, 4 waves each (dattn_core<PAIR>)
__device__ __forceinline__ void dstep_attn_down_body(const int* p_row_sp, const float* p_Q, const void* p_Kc, const void* p_Vc, long long p_slot_stride,
                                                     int p_max_ctx, int p_n_heads, float p_scale, int p_identity, int p_gemm_wgs, int p_M,
                                                     const DAttnArgs& a, const DGemmArgs& g, const DRowFin& fin) {
  constexpr int UNITS = NW / 2;
  __shared__ float red[NW][8][10];
  __shared__ float osh[PAIR ? 128 : 64];
  __shared__ __attribute__((aligned(16))) float sh[UNITS * 16 * 64 > 512 ? UNITS * 16 * 64 : 512];      // attention: 512 staged outputs (per row); GEMM: the units' cross-wave sums
  if constexpr (PAIR) {
    static_assert(!PAIR || (NW == 8 && !FIN), "paired rows: 8 waves, no in-launch row finish");
    if ((int)blockIdx.x >= p_gemm_wgs) {
      const int lid = blockIdx.x - p_gemm_wgs, P = (p_M + 1) >> 1, pr = lid % P, half = threadIdx.x >> 8;
      int m = 2 * pr + half;
      const bool dup = m >= p_M;
      m = dup ? 2 * pr : m;
      float (&redh)[4][8][10] = *reinterpret_cast<float (*)[4][8][10]>(&red[4 * half][0][0]);
      dattn_core<d16, true, 4, false, true>(m, lid / P, redh, osh + 64 * half, sh + 512 * half, p_row_sp, p_Q, p_Kc, p_Vc, p_slot_stride, p_max_ctx, p_n_heads, p_scale, p_identity, a,
                                             nullptr, osh + 64 * (half ^ 1), sh + 512 * (half ^ 1), dup);
      return;
    }
  }
  if ((int)blockIdx.x >= p_gemm_wgs) {
    const int lid = blockIdx.x - p_gemm_wgs;
    dattn_core<d16, true, NW, FIN>(lid % p_M, lid / p_M, red, osh, sh, p_row_sp, p_Q, p_Kc, p_Vc, p_slot_stride, p_max_ctx, p_n_heads, p_scale, p_identity, a, &fin);
    return;
  }
  // ---- GEMM role
  const int tid = threadIdx.x, unit = tid >> 7, tl = tid & 127, lane = tl & 63, wave = tl >> 6, r = lane & 31, h = lane >> 5;
  const int RT = (p_M + 31) / 32, FT = g.Npad / 32, KS = g.k_splits;
  const int bid = blockIdx.x, slot = (bid >> 3) * UNITS + unit, wt = (slot / RT) * 8 + (bid & 7);
  const bool valid = wt < FT * KS;
  const int wtc = valid ? wt : 0;
  const int bz = wtc / FT, m0 = (slot % RT) * 32, n0 = (wtc - bz * FT) * 32;
  int gm = m0 + r; gm = gm < p_M ? gm : p_M - 1;
  const int Kz = 512;                                  // one slab = 512 input columns (checked by the launcher)
  const int kb = bz * Kz + wave * 256;
  const d16* wrow = reinterpret_cast<const d16*>(g.W) + (long long)(n0 + r) * g.K + kb + h * 8;
  const d16* xrow = g.Xb + (long long)gm * g.ldx + kb + h * 8;
  // (the same 16 MFMAs in the same order as k_dgemm_s, but fed in four passes of 4 k-steps: this role shares the launch -- and
  // so the register allocation -- with the attention role; four short round trips of a few dozen workgroups hide behind the
  // attention workgroups)
  SS_DECL(); SS(0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    d16x8 wf[4], xf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { wf[s4] = ABL_GEMMW_LOAD(wrow + (ps * 4 + s4) * 16); xf[s4] = *reinterpret_cast<const d16x8*>(xrow + (ps * 4 + s4) * 16); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = mfma32(wf[s4], xf[s4], acc);
  }
  SS(1);
  float* redg = sh + unit * (16 * 64);
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) redg[i * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wave != 0 || !valid) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] += redg[i * 64 + lane];
  const int m = m0 + r;
  if constexpr (FIN) {
    // slab bz, rows of this tile: write-through 16-byte stores, drain, then lane r (h == 0) arrives at its row's counter; a unit that
    // happens to arrive last for some rows (rare: the attention workgroups outlive the units) finishes them one after the other
    if (m < p_M) {
      const drsrc_t ys = d_rsrc(g.Y, (long long)g.k_splits * p_M * 512 * 4);
      const int vo = ((bz * p_M + m) * 512 + n0 + 4 * h) * 4;       // the row differs per lane: it belongs in the VECTOR offset (a per-lane scalar offset becomes a 32-trip readfirstlane loop); < 2^31
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o4 = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), ys, vo + 32 * q, 0, D_SC1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SS(2); SS_FLUSH(2, 1);
    // both lane halves of a row have drained their stores before the row's lane adds: the wait above is wave-wide
    int old = -1;
    if (h == 0 && m < p_M) old = D_ARRIVE(fin.cnt + m);
    unsigned long long last = __ballot(old == fin.target - 1);
    while (last) {
      const int rl = __builtin_ctzll(last);
      last &= last - 1;
      row_finish(fin, m0 + rl, p_M, lane);
      if (lane == 0) D_CNT_RESET(fin.cnt + m0 + rl);
    }
  } else {
    if (m >= p_M) return;
    DGemmArgs b = g;
    b.Y = g.Y + (long long)bz * p_M * g.ldy;
    dgemm_epilogue<true, DEPI_PARTIAL>(b, acc, m, n0, h);
    SS(2); SS_FLUSH(2, 1);
  }
}

// The launch itself.  `a.stamp` (null in every production graph): device-side span of THIS launch -- first workgroup's start to last
// workgroup's end on s_memrealtime (100 MHz, one clock for the whole chip) -- which is what a rocprofv3 kernel trace reports and what HIP
// events cannot give inside a hipGraph replay.  Cheap by construction (a 1 800-workgroup launch must not queue 5 000 atomics on one
// address): block 0 stores the start; every workgroup folds its end into ONE of 64 slots with a fire-and-forget atomicMax; the slots
// alternate between two banks by launch parity (a.stamp_par = layer & 1: consecutive launches of an engine differ), and wave 0 of block 0
// of the NEXT stamped launch -- the previous one is complete by stream order -- takes the maximum of the other bank, adds end - start to
// the running sum and re-arms the bank.  The host folds the last launch in (etd_decoder_stats).
// Layout (u64): [0] sum of spans, [1] launches, [2 + bank] start, [8 + 64 bank + i] end slots.
template <int NW, bool FIN, bool PAIR = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(FIN ? ETD_FIN_OCC : (PAIR ? ETD_AD_OCC : AdOcc<NW>::lo), AdOcc<NW>::hi))) void k_dstep_attn_down(const int* p_row_sp, const float* p_Q, const void* p_Kc, const void* p_Vc, long long p_slot_stride,
                                                         int p_max_ctx, int p_n_heads, float p_scale, int p_identity, int p_gemm_wgs, int p_M,
                                                         DAttnArgs a, DGemmArgs g, DRowFin fin) {
  attn_stamp_begin(a.stamp, a.stamp_par, (int)blockIdx.x);
  dstep_attn_down_body<NW, FIN, PAIR>(p_row_sp, p_Q, p_Kc, p_Vc, p_slot_stride, p_max_ctx, p_n_heads, p_scale, p_identity, p_gemm_wgs, p_M, a, g, fin);
  attn_stamp_end(a.stamp, a.stamp_par, (int)blockIdx.x);
}

// waves per attention workgroup: 4, or ETD_AD_WAVES = 8 / 16 (measurement builds).  Measured on MI355X, round 2 ((history: 4ac2f57) tools/runs/r2_run1.sh):
// 54 rows x ctx 320, one engine: 0.197 / 0.230 / 0.238 ms per step at 4 / 8 / 16 waves, four engines 9.98 / 9.17 / 8.21
// engine-steps per ms; 128 rows x ctx 512: 0.356 / 0.349 / 0.419 ms; 128 rows x ctx 3.5 k: 1.290 / 1.343 / 1.362 ms.  Requesting a
// (row, head)'s whole context at once does NOT shorten the launch: its ~8 us of fixed cost are not the key loop's round trips.  Nor are
// they the second pass of dense-weight fragments in the tail: with all 16 requested before the merge (100 registers, 4 waves per SIMD)
// one engine steps in 0.199 instead of 0.198 ms and four engines reach 9.97-10.03 engine-steps per ms either way ((history: 4ac2f57) tools/runs/r2_run12.sh).
// The shipped library holds ONE form per case: 4-wave one-row workgroups and the 8-wave paired-rows form (both without the in-launch row finish).  The 8 / 16-wave
// one-row forms (ETD_AD_WAVES) and the row finish (ETD_ROWFIN, DRowFin) are measured dead ends kept for the record: built only with -DETD_EXPERIMENTS
// (ETD_EXTRA_FLAGS=-DETD_EXPERIMENTS python -m etude_amd.build --force), which `etd_has_experiments()` reports.
static int ad_waves(int M) {
  (void)M;
#ifdef ETD_EXPERIMENTS
  static const int env = getenv("ETD_AD_WAVES") ? atoi(getenv("ETD_AD_WAVES")) : 0;
  return (env == 8 || env == 16) ? env : 4;
#else
  return 4;
#endif
}
int launch_dstep_attn_down(const DAttnArgs& a, const DGemmArgs& g, const DRowFin* fin, hipStream_t st) {
#ifndef ETD_EXPERIMENTS
  if (fin) ETD_FAIL(ETD_EINVAL, "dstep_attn_down: the in-launch row finish is built with -DETD_EXPERIMENTS only");
#endif
  if (fin && (!fin->cnt || fin->target != a.n_heads + 16 * g.k_splits || fin->nslab != 12 || g.k_splits + a.n_heads != 12 || fin->P != g.Y ||
              a.dense_out != g.Y + (size_t)g.k_splits * a.M * 512 || !fin->bias || !fin->hin || !fin->hout || fin->hin == fin->hout || (fin->x1 && (!fin->x2 || !fin->g1 || !fin->b1 || !fin->g2 || !fin->b2))))
    ETD_FAIL(ETD_EINVAL, "dstep_attn_down: bad row-finish arguments");
  if (a.M < 1 || a.M > DS_STEP_MAX_ROWS || a.n_heads < 1 || !a.row_sp || !a.dense_w || !a.dense_out || a.max_ctx < 256 ||
      g.M != a.M || !g.Xb || !g.W || !g.Y || g.ldy != 512 || g.N != 512 || g.Npad != 512 || g.k_splits < 1 || g.k_splits * 512 > g.K || (g.K % 8) || a.n_heads * 64 != 512)
    ETD_FAIL(ETD_EINVAL, "dstep_attn_down: bad arguments");
  // two rows of a head per 8-wave attention workgroup, the head's dense slice read once per pair (ETD_AD_PAIR=0: always one row per 4-wave workgroup).
  // Bit-identical results; measured at the end of round 2 ((history: 4ac2f57) tools/runs/r2_run132.sh, r2_run133.sh): the launch 14.8 -> 13.9 us, one engine's step
  // 0.1985 -> 0.1894 ms, four engines 9.78 -> 10.03 engine-steps / ms, the job +0.3 % (within its spread)
  // -- at the headline's shape (54 rows x ~340 keys).  Whether it pays depends on rows, context and on how many engines share the chip: the host decides
  // per call (DAttnArgs::pair, api_dec.hip etd_decoder_step); ETD_AD_PAIR=0 / 1 force either form
  static const int pair_env = ETD_XENV("ETD_AD_PAIR") ? atoi(ETD_XENV("ETD_AD_PAIR")) : -1;
  const bool pair = (pair_env < 0 ? a.pair != 0 : pair_env > 0) && !fin && ad_waves(a.M) == 4;
  const int nw = pair ? 8 : ad_waves(a.M), units = nw / 2;
  const int RT = (a.M + 31) / 32, FT = g.Npad / 32;
  const int slots = ((FT * g.k_splits + 7) / 8) * RT;               // per XCD
  const int gemm_wgs = ((slots + units - 1) / units) * 8;
  ProfScope ps("k_dstep_attn_down", st, 2.0 * a.M * 512 * (512.0 * g.k_splits + 512.0), a.bytes_hint + 512.0 * (512.0 * g.k_splits + 512.0) * 2);
  const DRowFin f0 = fin ? *fin : DRowFin{};
#define ETD_AD_LAUNCH(NW_, FIN_) hipLaunchKernelGGL((k_dstep_attn_down<NW_, FIN_>), dim3(gemm_wgs + a.M * a.n_heads), dim3(64 * NW_), 0, st, a.row_sp, a.Q, a.Kc, a.Vc, a.slot_stride, a.max_ctx, a.n_heads, a.scale, \
                                              a.identity ? 1 : 0, gemm_wgs, a.M, a, g, f0)
  if (pair) {
    hipLaunchKernelGGL((k_dstep_attn_down<8, false, true>), dim3(gemm_wgs + ((a.M + 1) / 2) * a.n_heads), dim3(512), 0, st, a.row_sp, a.Q, a.Kc, a.Vc, a.slot_stride, a.max_ctx, a.n_heads, a.scale,
                       a.identity ? 1 : 0, gemm_wgs, a.M, a, g, f0);
  } else
#ifdef ETD_EXPERIMENTS
  if (fin) { if (nw == 16) ETD_AD_LAUNCH(16, true); else if (nw == 8) ETD_AD_LAUNCH(8, true); else ETD_AD_LAUNCH(4, true); }
  else if (nw == 16) ETD_AD_LAUNCH(16, false); else if (nw == 8) ETD_AD_LAUNCH(8, false); else ETD_AD_LAUNCH(4, false);
#else
  ETD_AD_LAUNCH(4, false);
#endif
#undef ETD_AD_LAUNCH
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

int launch_dattn(const DAttnArgs& a, bool kv_bf16, hipStream_t st) {
  if (a.M <= 0) ETD_FAIL(ETD_EINVAL, "dattn: bad M");
  ProfScope ps(a.M > 512 ? "k_dattn_prefill" : "k_dattn", st, 0, a.bytes_hint);
  dim3 g(a.M, a.n_heads);
  const int ident = a.identity && a.row_sp && a.max_ctx >= 64 ? 1 : 0;
  if (kv_bf16) hipLaunchKernelGGL(k_dattn<d16>, g, dim3(256), 0, st, a.row_sp, a.Q, a.Kc, a.Vc, a.slot_stride, a.max_ctx, a.n_heads, a.scale, ident, a);
  else hipLaunchKernelGGL(k_dattn<float>, g, dim3(256), 0, st, a.row_sp, a.Q, a.Kc, a.Vc, a.slot_stride, a.max_ctx, a.n_heads, a.scale, ident, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// LayerNorm of the residual stream for both parallel branches at once (input_layernorm and
// post_attention_layernorm read the same h, modeling_gpt_neox.py:250-270); one wave per row, d16 out.
__global__ __launch_bounds__(256) void k_ln_rows(const float* __restrict__ hsrc, int M, int H, const float* __restrict__ g1, const float* __restrict__ b1,
                                                 const float* __restrict__ g2, const float* __restrict__ b2, float eps, d16* __restrict__ x1, d16* __restrict__ x2) {
  // one wave per row, the row is read ONCE (8 floats per lane per 512 columns, H <= 2048) and kept in registers:
  // a kernel this small is pure latency, so one global round trip instead of three is the whole optimisation
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xp = hsrc + (long long)row * H;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < H) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xp + k), b = *reinterpret_cast<const f32x4*>(xp + k + 4);
      v[it][0] = a[0]; v[it][1] = a[1]; v[it][2] = a[2]; v[it][3] = a[3]; v[it][4] = b[0]; v[it][5] = b[1]; v[it][6] = b[2]; v[it][7] = b[3];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[it][j];
    }
  }
  s = wave_sum(s);
  const float mean = s / (float)H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
    if (lane * 8 + it * 512 < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d0 = v[it][j] - mean; q += d0 * d0; }
    }
  q = wave_sum(q);
  const float rstd = rsqrtf(q / (float)H + eps);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < H) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(g1 + k), gb = *reinterpret_cast<const f32x4*>(g1 + k + 4);
      const f32x4 ba = *reinterpret_cast<const f32x4*>(b1 + k), bb = *reinterpret_cast<const f32x4*>(b1 + k + 4);
      d16x8 o1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o1[j] = (d16)((v[it][j] - mean) * rstd * ga[j] + ba[j]); o1[4 + j] = (d16)((v[it][4 + j] - mean) * rstd * gb[j] + bb[j]); }
      *reinterpret_cast<d16x8*>(x1 + (long long)row * H + k) = o1;
      if (x2) {
        const f32x4 ha = *reinterpret_cast<const f32x4*>(g2 + k), hb = *reinterpret_cast<const f32x4*>(g2 + k + 4);
        const f32x4 ca = *reinterpret_cast<const f32x4*>(b2 + k), cb = *reinterpret_cast<const f32x4*>(b2 + k + 4);
        d16x8 o2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o2[j] = (d16)((v[it][j] - mean) * rstd * ha[j] + ca[j]); o2[4 + j] = (d16)((v[it][4 + j] - mean) * rstd * hb[j] + cb[j]); }
        *reinterpret_cast<d16x8*>(x2 + (long long)row * H + k) = o2;
      }
    }
  }
}
int launch_ln_rows(const float* hsrc, int M, int H, const float* g1, const float* b1, const float* g2, const float* b2, float eps,
                   d16* x1, d16* x2, hipStream_t st) {
  if (M <= 0 || H % 8 || H > 2048) ETD_FAIL(ETD_EINVAL, "ln_rows: bad shape");
  ProfScope ps("k_ln_rows", st, 0, (double)M * H * (4 + (x2 ? 4 : 2)));
  hipLaunchKernelGGL(k_ln_rows, dim3((M + 3) / 4), dim3(256), 0, st, hsrc, M, H, g1, b1, g2, b2, eps, x1, x2);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// hout = ((sum_z P[z] + bias) + add) + hin, then the next layer's two LayerNorms -> d16.  One wave per row, single pass.
#ifndef ETD_RL_ROWS
#define ETD_RL_ROWS 4      // rows (= waves) per workgroup of the row kernel
#endif
template <int KS>   // KS > 0: slab count known at compile time -> all slab loads of a row are in flight together (a runtime loop makes them dependent round trips)
__global__ __launch_bounds__(64 * ETD_RL_ROWS) void k_resid_ln_rows(const float* __restrict__ P, int ks_rt, const float* __restrict__ bias, const float* __restrict__ add,
                                                       const float* __restrict__ hin, float* __restrict__ hout, int M, int H,
                                                       const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2,
                                                       const float* __restrict__ b2, float eps, d16* __restrict__ x1, d16* __restrict__ x2) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * ETD_RL_ROWS + (threadIdx.x >> 6);
  if (row >= M) return;
  SS_DECL(); SS(0);
  const long long ro = (long long)row * H;
  float v[4][8];
  float s = 0.f;
  // the first 512 columns' LayerNorm parameters are requested together with the row itself: a kernel this small is one
  // global round trip long, a second dependent one (parameters after the statistics) would double it
  const int k0 = lane * 8;
  f32x4 pg1[2], pb1[2], pg2[2], pb2[2];
  if (x1 && k0 < H) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      pg1[hh] = *reinterpret_cast<const f32x4*>(g1 + k0 + 4 * hh); pb1[hh] = *reinterpret_cast<const f32x4*>(b1 + k0 + 4 * hh);
      pg2[hh] = *reinterpret_cast<const f32x4*>(g2 + k0 + 4 * hh); pb2[hh] = *reinterpret_cast<const f32x4*>(b2 + k0 + 4 * hh);
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < H) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      if constexpr (KS > 0) {
        f32x4 pa[KS], pb[KS];
#pragma unroll
        for (int z = 0; z < KS; ++z) {
          const float* pp = P + (long long)z * M * H + ro + k;
          pa[z] = *reinterpret_cast<const f32x4*>(pp); pb[z] = *reinterpret_cast<const f32x4*>(pp + 4);   // (a nontemporal hint here measured 2 % slower)
        }
#pragma unroll
        for (int z = 0; z < KS; ++z)
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += pa[z][j]; acc[4 + j] += pb[z][j]; }
      } else {
        for (int z = 0; z < ks_rt; ++z) {
          const float* pp = P + (long long)z * M * H + ro + k;
          const f32x4 a = *reinterpret_cast<const f32x4*>(pp), b = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += a[j]; acc[4 + j] += b[j]; }
        }
      }
      const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + k), bb = *reinterpret_cast<const f32x4*>(bias + k + 4);
      f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
      if (add) { da = *reinterpret_cast<const f32x4*>(add + ro + k); db = *reinterpret_cast<const f32x4*>(add + ro + k + 4); }
      const f32x4 ha = *reinterpret_cast<const f32x4*>(hin + ro + k), hb = *reinterpret_cast<const f32x4*>(hin + ro + k + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[it][j] = ((acc[j] + ba[j]) + da[j]) + ha[j]; v[it][4 + j] = ((acc[4 + j] + bb[j]) + db[j]) + hb[j]; }
      const f32x4 oa = {v[it][0], v[it][1], v[it][2], v[it][3]}, ob = {v[it][4], v[it][5], v[it][6], v[it][7]};
      *reinterpret_cast<f32x4*>(hout + ro + k) = oa;
      *reinterpret_cast<f32x4*>(hout + ro + k + 4) = ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[it][j];
    }
  }
  if (!x1) return;
  SS(1);                 // slabs, bias, residual summed (loads landed), row stored
  s = wave_sum(s);
  const float mean = s / (float)H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it)
    if (lane * 8 + it * 512 < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d0 = v[it][j] - mean; q += d0 * d0; }
    }
  q = wave_sum(q);
  const float rstd = rsqrtf(q / (float)H + eps);
  SS(2);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < H) {
      d16x8 o1, o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = (v[it][j] - mean) * rstd;
        const float w1 = it == 0 ? pg1[j >> 2][j & 3] : g1[k + j], c1 = it == 0 ? pb1[j >> 2][j & 3] : b1[k + j];
        const float w2 = it == 0 ? pg2[j >> 2][j & 3] : g2[k + j], c2 = it == 0 ? pb2[j >> 2][j & 3] : b2[k + j];
        o1[j] = (d16)(n * w1 + c1);
        o2[j] = (d16)(n * w2 + c2);
      }
      *reinterpret_cast<d16x8*>(x1 + ro + k) = o1;
      *reinterpret_cast<d16x8*>(x2 + ro + k) = o2;
    }
  }
  SS(3); SS_FLUSH(3, 0);
}
int launch_resid_ln_rows(const float* P, int k_splits, const float* bias, const float* add, const float* hin, float* hout, int M, int H,
                         const float* g1, const float* b1, const float* g2, const float* b2, float eps, d16* x1, d16* x2, hipStream_t st) {
  if (M <= 0 || H % 8 || H > 2048 || k_splits < 1) ETD_FAIL(ETD_EINVAL, "resid_ln_rows: bad shape");
  ProfScope ps("k_resid_ln_rows", st, 0, (double)M * H * 4 * (k_splits + 3));
  if (k_splits == 5) hipLaunchKernelGGL(k_resid_ln_rows<5>, dim3((M + ETD_RL_ROWS - 1) / ETD_RL_ROWS), dim3(64 * ETD_RL_ROWS), 0, st, P, k_splits, bias, add, hin, hout, M, H, g1, b1, g2, b2, eps, x1, x2);
  else if (k_splits == 12) hipLaunchKernelGGL(k_resid_ln_rows<12>, dim3((M + ETD_RL_ROWS - 1) / ETD_RL_ROWS), dim3(64 * ETD_RL_ROWS), 0, st, P, k_splits, bias, add, hin, hout, M, H, g1, b1, g2, b2, eps, x1, x2);
  else hipLaunchKernelGGL(k_resid_ln_rows<0>, dim3((M + ETD_RL_ROWS - 1) / ETD_RL_ROWS), dim3(64 * ETD_RL_ROWS), 0, st, P, k_splits, bias, add, hin, hout, M, H, g1, b1, g2, b2, eps, x1, x2);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, int n, int H, float* __restrict__ out) {
  const int i = blockIdx.x;
  const float* sp = src + (long long)idx[i] * H;
  for (int k = threadIdx.x * 4; k < H; k += blockDim.x * 4) *reinterpret_cast<f32x4*>(out + (long long)i * H + k) = *reinterpret_cast<const f32x4*>(sp + k);
}
int launch_gather_rows(const float* src, const int* idx, int n, int H, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_rows, dim3(n), dim3(128), 0, st, src, idx, n, H, out);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// embeddings: word + class + attribute_projection(cat(4 attribute embeddings))   etude_decoder.py:166-179
// the projection of each (attribute, bin) is precomputed at load: proj = bias + sum_a tab[a][bin_a]
// ================================================================================================
__global__ void k_dembed(DEmbedArgs a) {
  const int m = blockIdx.x;
  int id, cl, at[4];
  if (a.ids) {
    id = a.ids[m]; cl = a.cls[m];
#pragma unroll
    for (int k = 0; k < 4; ++k) at[k] = a.attrs[k * a.M + m];
  } else {
    // decode step: this kernel also materialises the row metadata the rest of the step reads (slot, position, active)
    const int slot = a.slots ? a.slots[m] : a.rows.slot[m];
    if (a.slots && threadIdx.x == 0) {
      const int ps = a.len[slot];
      a.row_slot_out[m] = slot; a.row_pos_out[m] = ps; a.row_active_out[m] = a.done[slot] ? 0 : 1;
      if (a.row_sp_out) { a.row_sp_out[2 * m] = slot; a.row_sp_out[2 * m + 1] = ps; }
    }
    id = a.cur_tok[slot]; cl = a.tgt_cls;
#pragma unroll
    for (int k = 0; k < 4; ++k) at[k] = a.tgt_attrs[slot * 4 + k];
  }
  for (int i = threadIdx.x; i < a.H; i += blockDim.x) {
    float p = a.attr_tab[(0 * a.n_bins + at[0]) * a.H + i];
    p += a.attr_tab[(1 * a.n_bins + at[1]) * a.H + i];
    p += a.attr_tab[(2 * a.n_bins + at[2]) * a.H + i];
    p += a.attr_tab[(3 * a.n_bins + at[3]) * a.H + i];
    a.h[(long long)m * a.H + i] = (a.word[(long long)id * a.H + i] + a.cls_emb[cl * a.H + i]) + p;
  }
}
int launch_dembed(const DEmbedArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_dembed, dim3(a.M), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// greedy argmax (lowest index on ties, like torch.argmax) + device-side stream state update
//                                                                    etude_decoder.py:333-343
// ================================================================================================
__global__ void k_dargmax(DArgmaxArgs a) {
  __shared__ float sp[256], ss[256];
  __shared__ int si[256];
  const int m = blockIdx.x, lane = threadIdx.x;
  const float* lg = a.logits + (long long)m * a.ldl;
  const int slot_r = a.rows.slot[m];
  const float inv_temp = a.samp ? a.samp->inv_temp : 0.f;
  int bi;
  if (inv_temp > 0.f) bi = wave_sample(lg, a.V, lane, inv_temp, a.samp->top_p, a.samp->seed, a.rng_key[slot_r], (unsigned)a.n_out[slot_r], sp, ss, si);
  else bi = wave_argmax(lg, a.V, lane);
  if (lane == 0 && a.rows.active[m]) {
    const int slot = slot_r;
    if (!a.done[slot]) {
      const int n = a.n_out[slot];
      if (n < a.out_cap) a.out_tok[(long long)slot * a.out_cap + n] = bi;
      a.n_out[slot] = n + 1;
      a.cur_tok[slot] = bi;
      a.len[slot] = a.rows.pos[m] + 1;
      if (bi == a.eos[slot] || n + 1 >= a.limit[slot]) a.done[slot] = 1;
    }
  }
}
int launch_dargmax(const DArgmaxArgs& a, hipStream_t st) {
  if (a.samp && (a.V > 256 || !a.rng_key)) ETD_FAIL(ETD_EINVAL, "dargmax: sampling needs a vocabulary <= 256 and stream keys");
  hipLaunchKernelGGL(k_dargmax, dim3(a.M), dim3(64), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

