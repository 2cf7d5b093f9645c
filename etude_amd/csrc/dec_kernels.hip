// EtudeDecoder (GPT-NeoX, 8 x [LN, QKV+RoPE, causal attention, dense | LN, MLP(GELU)] parallel residual)
// kernels for gfx950.  Reference: etude/models/etude_decoder.py:148-206 and HF GPTNeoXLayer /
// GPTNeoXAttention (transformers modeling_gpt_neox.py:195-281, RoPE :111-151).
//
// Two weight precisions share every kernel:
//   * fp32 ("parity mode"): exact-fp32 products on v_mfma_f32_32x32x2_f32 (bit-identical to an fmaf
//     chain), fp32 KV cache -- this is what the greedy token-id parity gate runs on;
//   * bf16: v_mfma_f32_32x32x16_bf16 on bf16 weights and a bf16 KV cache (the HBM-bound serving mode).
// The residual stream, LayerNorm, RoPE, softmax and the logits are fp32 in both.
#include "dec_kernels.h"
#include "prof.h"

#define DLD32 36   // fp32 LDS row stride (floats) for a 32-wide K chunk: 144 B
#define DLD16 72   // bf16 LDS row stride (elements) for a 64-wide K chunk: 144 B

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

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
    bf16* Xs = reinterpret_cast<bf16*>(smem);
    bf16* Ws = Xs + 32 * DLD16;
    const bf16* W = reinterpret_cast<const bf16*>(a.W) + (long long)n0 * a.K;
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      {  // X: 32 rows x 8 chunks of 8 floats -> bf16
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
        bf16x8 o = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3], (bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
        *reinterpret_cast<bf16x8*>(Xs + row * DLD16 + ch * 8) = o;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // W: 128 rows x 8 chunks of 16 B
        const int c = tid + i * 256, row = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(Ws + row * DLD16 + ch * 8) = *reinterpret_cast<const u32x4*>(W + (long long)row * a.K + k0 + ch * 8);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + (wave * 32 + r) * DLD16 + s * 16 + h * 8);
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xs + r * DLD16 + s * 16 + h * 8);
        acc = mfma32(wf, xf, acc);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane = token m, registers = features nb + acc_row(i, h)
  const int m = m0 + r;
  if (m >= a.M) return;
  const int nb = n0 + wave * 32;
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
    if (part == 0) {
      float* qp = a.Q + (long long)m * (a.n_heads * 64) + head * 64 + dbase;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; *reinterpret_cast<f32x4*>(qp + 8 * q + 4 * h) = o; }
    } else if (a.rows.active[m] && pos < a.max_ctx) {
      const long long off = (long long)a.rows.slot[m] * a.slot_stride + ((long long)head * a.max_ctx + pos) * 64 + dbase;
      void* base = part == 1 ? a.Kc : a.Vc;
      if constexpr (WBF16) {
        bf16* kp = reinterpret_cast<bf16*>(base) + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x4*>(kp + 8 * q + 4 * h) = pack4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
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
        if (EPI != DEPI_LOGITS) v[j] += a.bias[n + j];
        if (EPI == DEPI_GELU) v[j] = gelu_erf(v[j]);
      }
      if constexpr (EPI == DEPI_RESID) {
        const f32x4 ad = *reinterpret_cast<const f32x4*>(a.add + (long long)m * a.N + n);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(a.hin + (long long)m * a.N + n);
        const f32x4 o = {(v[0] + ad[0]) + hi[0], (v[1] + ad[1]) + hi[1], (v[2] + ad[2]) + hi[2], (v[3] + ad[3]) + hi[3]};
        *reinterpret_cast<f32x4*>(a.hout + (long long)m * a.N + n) = o;
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

template <bool WBF16>
static void dgemm_dispatch(const DGemmArgs& a, int epi, dim3 g, hipStream_t st) {
  switch (epi) {
    case DEPI_BIAS: hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_BIAS>), g, dim3(256), 0, st, a); break;
    case DEPI_GELU: hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_GELU>), g, dim3(256), 0, st, a); break;
    case DEPI_RESID: hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_RESID>), g, dim3(256), 0, st, a); break;
    case DEPI_LOGITS: hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_LOGITS>), g, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_dgemm<WBF16, DEPI_QKV>), g, dim3(256), 0, st, a); break;
  }
}

int launch_dgemm(const DGemmArgs& a, int epi, bool w_bf16, hipStream_t st) {
  if (a.M <= 0 || a.Npad % 128 || a.K % 64 || a.N > a.Npad) ETD_FAIL(ETD_EINVAL, "dgemm: bad shape M=%d N=%d Npad=%d K=%d", a.M, a.N, a.Npad, a.K);
  if (a.ln_g && a.K % 256) ETD_FAIL(ETD_EINVAL, "dgemm: LayerNorm prologue needs K %% 256 == 0");
  if (epi == DEPI_QKV && (a.rot_half != 8 || a.N % 192)) ETD_FAIL(ETD_EINVAL, "dgemm: QKV epilogue needs head_dim 64 and rotary_ndims 16");
  if (epi == DEPI_RESID && (a.N % 4)) ETD_FAIL(ETD_EINVAL, "dgemm: resid needs N %% 4 == 0");
  ProfScope ps("k_dgemm", st, 2.0 * a.M * a.N * a.K, (double)a.Npad * a.K * (w_bf16 ? 2 : 4));
  dim3 g((a.M + 31) / 32, a.Npad / 128);
  if (w_bf16) dgemm_dispatch<true>(a, epi, g, st);
  else dgemm_dispatch<false>(a, epi, g, st);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_dattn: causal attention of each row's query against its slot's KV cache [0, pos]
//                                    modeling_gpt_neox.py:172-190 (softmax in fp32), :222-236
// grid (M, heads); 4 waves split the key blocks; a wave-iteration covers 8 keys: lane = (key j = lane>>3,
// 8-dim chunk c = lane&7), so K/V loads are fully coalesced 2 KB (fp32) / 1 KB (bf16) blocks.
// ================================================================================================
template <typename KVT> __device__ __forceinline__ void load8(const KVT* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = bf2f(a[j]);
}

template <typename KVT>
__global__ __launch_bounds__(256) void k_dattn(DAttnArgs a) {
  __shared__ float red[4][8][10];            // per wave, per dim-chunk: m, l, o[8]
  const int m = blockIdx.x, head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 3, c = lane & 7;
  const int slot = a.rows.slot[m], pos = a.rows.pos[m];
  const int ctx = (pos < a.max_ctx ? pos : a.max_ctx - 1) + 1;
  const int hidden = a.n_heads * 64;
  float q[8];
  {
    const float* qp = a.Q + (long long)m * hidden + head * 64 + c * 8;
    const f32x4 x = *reinterpret_cast<const f32x4*>(qp), y = *reinterpret_cast<const f32x4*>(qp + 4);
    q[0] = x[0] * a.scale; q[1] = x[1] * a.scale; q[2] = x[2] * a.scale; q[3] = x[3] * a.scale;
    q[4] = y[0] * a.scale; q[5] = y[1] * a.scale; q[6] = y[2] * a.scale; q[7] = y[3] * a.scale;
  }
  const KVT* kb = reinterpret_cast<const KVT*>(a.Kc) + (long long)slot * a.slot_stride + (long long)head * a.max_ctx * 64;
  const KVT* vb = reinterpret_cast<const KVT*>(a.Vc) + (long long)slot * a.slot_stride + (long long)head * a.max_ctx * 64;
  float mr = -INFINITY, lr = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int k0 = wave * 8; k0 < ctx; k0 += 32) {
    const int key = k0 + j;
    const bool valid = key < ctx;
    const int kk = valid ? key : ctx - 1;
    float kv[8];
    load8<KVT>(kb + (long long)kk * 64 + c * 8, kv);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(q[e], kv[e], s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    load8<KVT>(vb + (long long)kk * 64 + c * 8, kv);
    if (valid) {
      const float mn = fmaxf(mr, s);
      const float al = expf(mr - mn), p = expf(s - mn);
      lr = lr * al + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * al + p * kv[e];
      mr = mn;
    }
  }
  // merge the 8 key slots of this wave (lanes differing in bits 3..5), then the 4 waves through LDS
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(mr, off, 64), l2 = __shfl_xor(lr, off, 64);
    const float mn = fmaxf(mr, m2);
    const float f1 = (mr == -INFINITY) ? 0.f : expf(mr - mn), f2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    lr = lr * f1 + l2 * f2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float o2 = __shfl_xor(o[e], off, 64); o[e] = o[e] * f1 + o2 * f2; }
    mr = mn;
  }
  if (j == 0) {
    red[wave][c][0] = mr; red[wave][c][1] = lr;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c][2 + e] = o[e];
  }
  __syncthreads();
  if (wave == 0 && j == 0) {
    float M2 = -INFINITY;
    for (int w = 0; w < 4; ++w) M2 = fmaxf(M2, red[w][c][0]);
    float L = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) {
      const float mw = red[w][c][0];
      const float f = (mw == -INFINITY) ? 0.f : expf(mw - M2);
      L += red[w][c][1] * f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += red[w][c][2 + e] * f;
    }
    const float inv = 1.f / L;
    float* op = a.O + (long long)m * hidden + head * 64 + c * 8;
    const f32x4 x = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv}, y = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
    *reinterpret_cast<f32x4*>(op) = x;
    *reinterpret_cast<f32x4*>(op + 4) = y;
  }
}

int launch_dattn(const DAttnArgs& a, bool kv_bf16, hipStream_t st) {
  if (a.M <= 0) ETD_FAIL(ETD_EINVAL, "dattn: bad M");
  ProfScope ps("k_dattn", st, 0, 0);
  dim3 g(a.M, a.n_heads);
  if (kv_bf16) hipLaunchKernelGGL(k_dattn<bf16>, g, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_dattn<float>, g, dim3(256), 0, st, a);
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
    const int slot = a.rows.slot[m];
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
  const int m = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int v = lane; v < a.V; v += 64) {
    const float x = a.logits[(long long)m * a.ldl + v];
    if (x > best || (x == best && v < bi)) { best = x; bi = v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0 && a.rows.active[m]) {
    const int slot = a.rows.slot[m];
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
  hipLaunchKernelGGL(k_dargmax, dim3(a.M), dim3(64), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

__global__ void k_decode_rows(const int* slots, int n, const int* len, const int* done, int* row_slot, int* row_pos, int* row_active) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int s = slots[i]; row_slot[i] = s; row_pos[i] = len[s]; row_active[i] = done[s] ? 0 : 1; }
}
int launch_decode_rows(const int* slots_dev, int n, const int* len, const int* done, int* row_slot, int* row_pos, int* row_active, hipStream_t st) {
  hipLaunchKernelGGL(k_decode_rows, dim3((n + 63) / 64), dim3(64), 0, st, slots_dev, n, len, done, row_slot, row_pos, row_active);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
