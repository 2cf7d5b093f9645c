// Fused hFT-Transformer sub-layers for gfx950: the token tile stays in REGISTERS between the GEMMs of a sub-layer, the
// weights stream through an LDS ring by LDS-DMA (global_load_lds), and only the sub-layer's input and output touch HBM.
//
// Round 1 ran every Linear as its own launch: per encoder layer and window 1.2 GB of HBM traffic for 67 MB of layer input
// (profiles/r01_profile_summary.txt), and its GEMMs sat at the HBM's 4.3 TB/s of mixed traffic instead of on the MFMA pipes.
//
// k_ffn_fused:  Y = LayerNorm(X + relu(X W1^T + b1) W2^T + b2) * gamma + beta       amt_apc.py:250-259, 383-392
//   * one wave owns 32 tokens (the MFMA's lane dimension) for the whole sub-layer; a workgroup is 4 such waves, two workgroups
//     share a CU (<= 256 registers, 2 x 69 KiB of LDS);
//   * X enters once as the B-operand fragments of a 32x32x16 MFMA (16 fragments = 64 registers per lane) -- they serve the
//     first GEMM 16 times over and, at the end, the residual;
//   * the 512-wide hidden layer is produced 32 features at a time: acc1 = W1[32 rows] . X  (16 dependent MFMAs), bias + ReLU +
//     bf16 rounding in registers, and that accumulator IS the B operand of the second GEMM (guide section 3, "an accumulator
//     tile as the next MFMA's operand": the k order inside a step is permuted, so W2's fragments are packed on the host in that
//     order) -- the hidden activations never exist anywhere but in 8 registers;
//   * acc2 = 8 tiles x 16 registers holds the token's 256 outputs; W2's output rows are permuted on the host so that register
//     (tile t, i) of lane half h is feature 32 t + 16 (i >> 3) + 8 h + (i & 7): exactly the layout of the X fragments, so the
//     residual is a register-to-register add and the normalised row leaves as 16-byte pieces;
//   * weights: fragment order in global memory (1 KiB per (tile, k-step), lane l's 16 bytes at offset 16 l), so an LDS-DMA
//     piece is a plain copy, every ds_read_b128 of a fragment is conflict-free and needs no address arithmetic; 32 KiB per
//     32 hidden features, two ring slots, one barrier per slot.
#include "ext_kernels.h"
#include "prof.h"

#define FFN_SLOT_ELEMS (16 * 1024)      // e16 elements per ring slot: 32 fragments of 512 elements (1 KiB)
#define FFN_NSUB 16                     // 512 hidden features / 32

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256, 2) void k_ffn_fused(FfnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FFN_SLOT_ELEMS * 2 + (512 + 3 * 256) * 4];
  e16* ring = reinterpret_cast<e16*>(smem);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * FFN_SLOT_ELEMS * 2);      // b1[512] | b2[256] | gamma[256] | beta[256]
  float* sb2 = sb1 + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128, m = m0 + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;

  // ---- ring slot `sl` <- the 32 KiB of sub-chunk `sc`: 32 one-KiB pieces, 8 per wave
  auto issue = [&](int sc, int sl) {
    const e16* src = a.Wf + (long long)sc * FFN_SLOT_ELEMS + wave * (8 * 512) + lane * 8;
    e16* dst = ring + sl * FFN_SLOT_ELEMS + wave * (8 * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  };
  issue(0, 0);
  // the token tile as B fragments: lane (token r, half h) holds X[token][16 s + 8 h .. + 8] for s = 0 .. 15
  e16x8 xf[16];
  {
    const e16* xp = a.X + (long long)mc * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const e16x8*>(xp + 16 * s);
  }
  for (int i = tid; i < 512; i += 256) sb1[i] = a.b1[i];
  sb2[tid] = a.b2[tid]; sb2[256 + tid] = a.gamma[tid]; sb2[512 + tid] = a.beta[tid];

  f32x16 acc2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;

  for (int sc = 0; sc < FFN_NSUB; ++sc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of slot sc & 1 have landed ...
    __syncthreads();                                          // ... and everybody's; every wave is done reading the other slot
    if (sc + 1 < FFN_NSUB) issue(sc + 1, (sc + 1) & 1);
    const e16* sl = ring + (sc & 1) * FFN_SLOT_ELEMS + lane * 8;
    // hidden features 32 sc .. + 32 of the wave's 32 tokens
    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
    // (fragments are requested four at a time, one group ahead of the MFMAs that use them: left alone, hipcc hoists all 32 reads of
    // a sub-chunk to the top and spills; the other workgroup's wave on this SIMD covers the LDS latency of a group)
    e16x8 af[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + k * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) af[(g + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + ((g + 1) * 4 + k) * 512);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + (16 + k) * 512);      // first four W2 fragments (ks 0, tiles 0..3)
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc1 = mfma32(af[g & 1][k], xf[g * 4 + k], acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
    e16x8 hf[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + 32 * sc + 8 * q + 4 * h);       // register 4 q + j is hidden feature 32 sc + 8 q + 4 h + j
#pragma unroll
      for (int j = 0; j < 4; ++j) hf[q >> 1][4 * (q & 1) + j] = (e16)fmaxf(acc1[4 * q + j] + bb[j], 0.f);
    }
    // out[256] += W2[:, these 32 hidden features] . hidden   (two 16-deep k-steps, 8 output tiles)
#pragma unroll
    for (int g = 0; g < 4; ++g) {                              // group g = fragments 16 + 4 g .. + 4: k-step g >> 1, tiles 4 (g & 1) .. + 4
      if (g < 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) af[(g + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + (16 + (g + 1) * 4 + k) * 512);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc2[4 * (g & 1) + k] = mfma32(af[g & 1][k], hf[g >> 1], acc2[4 * (g & 1) + k]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- bias + residual + LayerNorm over the token's 256 features (128 in this lane, 128 in lane ^ 32)
  float s1 = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f0 = 32 * t + 16 * u + 8 * h + 4 * q;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(sb2 + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc2[t][8 * u + 4 * q + j] + bb[j] + bf2f(xf[2 * t + u][4 * q + j]);
          acc2[t][8 * u + 4 * q + j] = v;
          s1 += v;
        }
      }
  s1 += xhalf(s1);
  const float mean = s1 * (1.f / 256.f);
  float s2 = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = acc2[t][i] - mean; s2 += d * d; }
  s2 += xhalf(s2);
  const float rstd = rsqrtf(s2 * (1.f / 256.f) + 1e-5f);
  e16* yp = a.Y + (long long)m * 256 + 8 * h;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      e16x8 o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f0 = 32 * t + 16 * u + 8 * h + 4 * q;
        const f32x4 gg = *reinterpret_cast<const f32x4*>(sb2 + 256 + f0), be = *reinterpret_cast<const f32x4*>(sb2 + 512 + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * q + j] = (e16)((acc2[t][8 * u + 4 * q + j] - mean) * rstd * gg[j] + be[j]);
      }
      if (m < a.M) *reinterpret_cast<e16x8*>(yp + 16 * (2 * t + u)) = o;
    }
}

int launch_ffn_fused(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0 || !a.X || !a.Wf || !a.b1 || !a.b2 || !a.gamma || !a.beta || !a.Y || (((uintptr_t)a.X | (uintptr_t)a.Y | (uintptr_t)a.Wf) & 15))
    ETD_FAIL(ETD_EINVAL, "ffn_fused: bad arguments");
  ETD_LAUNCH_FILTER("k_ffn_fused");
  ProfScope ps("k_ffn_fused", st, 2.0 * a.M * 256.0 * 512.0 * 2.0, (double)a.M * 256 * 2 * 2 + 512.0 * 256 * 2 * 2);
  hipLaunchKernelGGL(k_ffn_fused, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// Host side: fc_1 [512][256] and fc_2 [256][512] (fp32, nn.Linear layout) -> the kernel's weight stream, e16:
// [sub-chunk sc of 32 hidden features][32 fragments][lane 64][8]; fragments 0..15 = W1 (k-step s), 16..31 = W2 (k-step ks, tile t).
void pack_ffn_weights(const float* W1, const float* W2, uint16_t* dst, uint16_t (*f2bf)(float)) {
  for (int sc = 0; sc < FFN_NSUB; ++sc)
    for (int fr = 0; fr < 32; ++fr)
      for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        uint16_t* d = dst + (((size_t)sc * 32 + fr) * 64 + l) * 8;
        if (fr < 16) {
          const int s = fr;                                    // A[row = hidden 32 sc + r][k = 16 s + 8 h + j]
          for (int j = 0; j < 8; ++j) d[j] = f2bf(W1[(size_t)(32 * sc + r) * 256 + 16 * s + 8 * h + j]);
        } else {
          const int ks = (fr - 16) >> 3, t = (fr - 16) & 7;
          // output row held by A-row r of tile t: accumulator register i = (r & 3) + 4 (r >> 3) of lane half (r >> 2) & 1 must be
          // feature 32 t + 16 (i >> 3) + 8 half + (i & 7) -- the layout of the X fragments
          const int i = (r & 3) + 4 * (r >> 3), hh = (r >> 2) & 1;
          const int feat = 32 * t + 16 * (i >> 3) + 8 * hh + (i & 7);
          // k order of an accumulator used as B operand: element j of lane half h is row 16 ks + 8 (j >> 2) + 4 h + (j & 3)
          for (int j = 0; j < 8; ++j) d[j] = f2bf(W2[(size_t)feat * 512 + 32 * sc + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)]);
        }
      }
}


// ================================================================================================
// k_proj256: every K = 256 projection of the model on the same skeleton -- the wave's 32 tokens as B fragments in registers,
// weights from the LDS ring, one 256-feature output block after another for the SAME token tile (X is read once for Q, K and V,
// once for the three decoder layers' cross-attention K/V ...).  Block kinds:
//   ROW  bias (+ReLU) -> bf16 row-major                                                     amt_apc.py:342-343 (fc_q, fc_k)
//   VT   bias -> V^T[(seq, head)][d][pos] for k_attn: the MFMA is issued the other way round (A = tokens, B = weights), the
//        accumulator then has the feature on the lane; pairs of 8-byte token groups are exchanged between the lane halves
//        (v_permlane32_swap) and leave as 16-byte pieces                                     amt_apc.py:344 (fc_v)
//   LN   bias + residual + LayerNorm (the layer's shared one) -> bf16 row-major            amt_apc.py:371 (fc_o), :250
// Weight stream per block: [chunk c of 4 k-steps][k-step kk][tile t][lane][8] = 4 x 32 KiB; ROW / LN blocks have their output
// rows permuted like k_ffn_fused's second GEMM (accumulator layout == fragment layout), VT blocks are in natural order.
// ================================================================================================
// one 256-feature block for the wave's 32 tokens; the ring keeps running across blocks (chunk g + 1 may belong to the next block)
template <int KIND>
__device__ __forceinline__ void proj_block(const ProjArgs& a, const ProjBlock pb, const ProjBlock* sblk, const int b, const int nchunk, e16* ring,
                                           const e16x8 (&xf)[16], const int wave, const int lane, const int m0w, float* sbias) {
  const int r = lane & 31, h = lane >> 5, m = m0w + r;
  const int mc = m < a.M ? m : a.M - 1;
  // this block's bias (and the LayerNorm parameters) go to LDS now; the K loop's barriers publish them long before the epilogue
  // (read from global memory in the epilogue, hipcc hoists ~50 loads above it and spills the accumulators)
  sbias[threadIdx.x] = pb.bias[threadIdx.x];
  if constexpr (KIND == PROJ_LN) { sbias[256 + threadIdx.x] = a.gamma[threadIdx.x]; sbias[512 + threadIdx.x] = a.beta[threadIdx.x]; }
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int g = b * 4 + c;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (g + 1 < nchunk) {
      const e16* src = sblk[(g + 1) >> 2].Wf + (long long)((g + 1) & 3) * FFN_SLOT_ELEMS + wave * (8 * 512) + lane * 8;
      e16* dst = ring + ((g + 1) & 1) * FFN_SLOT_ELEMS + wave * (8 * 512);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
    }
    const e16* sl = ring + (g & 1) * FFN_SLOT_ELEMS + lane * 8;
    e16x8 af[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + k * 512);
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {                   // group gq = fragments 4 gq .. + 4: k-step gq >> 1, tiles 4 (gq & 1) .. + 4
      if (gq < 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) af[(gq + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + ((gq + 1) * 4 + k) * 512);
      }
      const e16x8 xb = xf[4 * c + (gq >> 1)];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (KIND == PROJ_VT || KIND == PROJ_VFRAG) acc[4 * (gq & 1) + k] = mfma32(xb, af[gq & 1][k], acc[4 * (gq & 1) + k]);
        else                           acc[4 * (gq & 1) + k] = mfma32(af[gq & 1][k], xb, acc[4 * (gq & 1) + k]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue
  if constexpr (KIND == PROJ_KFRAG || KIND == PROJ_VFRAG) {
    // K / V as the MFMA-fragment images k_attn_frag streams: per (sequence, head) and 64-key step 16 KiB =
    // [K: key tile 2][k-step 4][lane][8] | [V: key tile 2][ks 2][dt 2][lane][8].  The wave's 32 tokens are ONE key tile, and the
    // accumulators already hold fragments: K (token on the lane, rows permuted) registers 8 u .. + 8 of tile t = k-step 2 (t & 1) + u
    // of head t >> 1; V (feature on the lane) registers 8 ks .. + 8 = k-step ks of d-tile t & 1.  Every store is 16 bytes per lane
    // at lane-linear addresses: one contiguous KiB per instruction.
    const int seq = m0w / a.S, kt = (m0w - seq * a.S) >> 5, step = kt >> 1, kt2 = kt & 1;
    if (m0w < a.M) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        e16* img = pb.dst + ((long long)(seq * 4 + (t >> 1)) * a.kv_nstep + step) * 8192 + lane * 8;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          e16x8 fr;
          if constexpr (KIND == PROJ_KFRAG) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + 32 * t + 16 * u + 8 * h + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) fr[4 * q + j] = (e16)(acc[t][8 * u + 4 * q + j] + bb[j]);
            }
            *reinterpret_cast<e16x8*>(img + (kt2 * 4 + 2 * (t & 1) + u) * 512) = fr;
          } else {
            const float bv = sbias[32 * t + r];
#pragma unroll
            for (int j = 0; j < 8; ++j) fr[j] = (e16)(acc[t][8 * u + j] + bv);
            *reinterpret_cast<e16x8*>(img + 4096 + ((kt2 * 2 + u) * 2 + (t & 1)) * 512) = fr;
          }
        }
      }
    }
  } else if constexpr (KIND == PROJ_VT) {
    // acc[t][i]: token (i & 3) + 8 (i >> 2) + 4 h of the wave's 32, feature 32 t + r
    const int seq = m0w / a.S, pos0 = m0w - seq * a.S;
    const bool vec = (a.S % 32 == 0) && (a.Spad % 8 == 0) && (m0w + 32 <= a.M);
    // sequence length a multiple of 4 but not of 32 (the 88-note self-attention of the frequency decoder): a lane's four tokens 8 q + 4 h + j of a
    // group q are consecutive positions of ONE sequence, so they leave as one 8-byte store; (sequence, position) of the four groups once per block
    const bool quad = !vec && (a.S % 4 == 0) && (a.Spad % 4 == 0) && (m0w + 32 <= a.M);
    long long qoff[4];
    if (quad) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mt = m0w + 8 * q + 4 * h, sq = mt / a.S;
        qoff[q] = (long long)sq * 4 * 64 * a.Spad + (mt - sq * a.S);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int f = 32 * t + r;
      const float bv = sbias[f];
      e16* row = pb.dst + ((long long)(seq * 4 + (f >> 6)) * 64 + (f & 63)) * a.Spad;
      if (vec) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const e16x4 lo = pack4e(acc[t][8 * p] + bv, acc[t][8 * p + 1] + bv, acc[t][8 * p + 2] + bv, acc[t][8 * p + 3] + bv);          // token group q = 2 p
          const e16x4 hi = pack4e(acc[t][8 * p + 4] + bv, acc[t][8 * p + 5] + bv, acc[t][8 * p + 6] + bv, acc[t][8 * p + 7] + bv);      // q = 2 p + 1
          const u32x2 la = __builtin_bit_cast(u32x2, lo), lb = __builtin_bit_cast(u32x2, hi);
          // lower half keeps its q = 2 p group and takes the upper half's; the upper half takes the lower's q = 2 p + 1 and keeps its own
          const auto s0 = __builtin_amdgcn_permlane32_swap(la[0], lb[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(la[1], lb[1], false, false);
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(row + pos0 + 16 * p + 8 * h) = o;
        }
      } else if (quad) {
        e16* base = pb.dst + ((long long)(f >> 6) * 64 + (f & 63)) * a.Spad;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<e16x4*>(base + qoff[q]) = pack4e(acc[t][4 * q] + bv, acc[t][4 * q + 1] + bv, acc[t][4 * q + 2] + bv, acc[t][4 * q + 3] + bv);
      } else {
        // (any other sequence length, or the ragged last tile)
        float vals[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) vals[i] = acc[t][i] + bv;
#pragma unroll 1
        for (int q = 0; q < 4; ++q)
#pragma unroll 1
          for (int j = 0; j < 4; ++j) {
            const int mt = m0w + 8 * q + 4 * h + j;
            if (mt < a.M) {
              const int sq = mt / a.S, ps = mt - sq * a.S;
              float v = vals[0];
#pragma unroll
              for (int i = 1; i < 16; ++i) v = (i == 4 * q + j) ? vals[i] : v;
              pb.dst[((long long)(sq * 4 + (f >> 6)) * 64 + (f & 63)) * a.Spad + ps] = (e16)v;
            }
          }
      }
    }
  } else if constexpr (KIND == PROJ_ROW) {
    e16* yp = pb.dst + (long long)m * pb.ldd + 8 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        e16x8 o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + 32 * t + 16 * u + 8 * h + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) { float v = acc[t][8 * u + 4 * q + j] + bb[j]; if (pb.relu) v = fmaxf(v, 0.f); o[4 * q + j] = (e16)v; }
        }
        if (m < a.M) *reinterpret_cast<e16x8*>(yp + 16 * (2 * t + u)) = o;
      }
  } else {
    // LN: the residual row arrives in pieces of the accumulators' own layout
    const int rrow = a.r_mod > 0 ? mc % a.r_mod : mc;
    const e16* rp = a.R + (long long)rrow * 256 + 8 * h;
    // (an LN block is the last block of its launch -- launch_proj256 checks -- so the X fragments are dead and the residual's 16
    // fragments take their registers: one round trip for the whole row)
    e16x8 rf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) rf[s] = *reinterpret_cast<const e16x8*>(rp + 16 * s);
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + 32 * t + 16 * u + 8 * h + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[t][8 * u + 4 * q + j] + bb[j] + bf2f(rf[2 * t + u][4 * q + j]);
            acc[t][8 * u + 4 * q + j] = v;
            s1 += v;
          }
        }
      }
    s1 += xhalf(s1);
    const float mean = s1 * (1.f / 256.f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = acc[t][i] - mean; s2 += d * d; }
    s2 += xhalf(s2);
    const float rstd = rsqrtf(s2 * (1.f / 256.f) + 1e-5f);
    e16* yp = pb.dst + (long long)m * pb.ldd + 8 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        e16x8 o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int f0 = 32 * t + 16 * u + 8 * h + 4 * q;
          const f32x4 gg = *reinterpret_cast<const f32x4*>(sbias + 256 + f0), be = *reinterpret_cast<const f32x4*>(sbias + 512 + f0);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * q + j] = (e16)((acc[t][8 * u + 4 * q + j] - mean) * rstd * gg[j] + be[j]);
        }
        if (m < a.M) *reinterpret_cast<e16x8*>(yp + 16 * (2 * t + u)) = o;
      }
  }
}

template <bool LNK>      // LNK: the launch is ONE LayerNorm block (its own instantiation: the X fragments die with the K loop)
__global__ __launch_bounds__(256, 2) void k_proj256(ProjArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FFN_SLOT_ELEMS * 2 + 2 * 768 * 4 + PROJ_MAX_BLOCKS * sizeof(ProjBlock)];
  e16* ring = reinterpret_cast<e16*>(smem);
  float* sbias = reinterpret_cast<float*>(smem + 2 * FFN_SLOT_ELEMS * 2);             // bias (| gamma | beta), two copies alternating by block
  ProjBlock* sblk = reinterpret_cast<ProjBlock*>(smem + 2 * FFN_SLOT_ELEMS * 2 + 2 * 768 * 4);      // the block list, indexable at run time (a by-value kernel argument is not)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int m0w = blockIdx.x * 128 + wave * 32, m = m0w + (lane & 31);
  const int mc = m < a.M ? m : a.M - 1;
  const int nchunk = a.nblk * 4;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < PROJ_MAX_BLOCKS; ++i) sblk[i] = a.blk[i];
  }
  {
    const e16* src = a.blk[0].Wf + wave * (8 * 512) + lane * 8;
    e16* dst = ring + wave * (8 * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  }
  e16x8 xf[16];
  {
    const e16* xp = a.X + (long long)mc * a.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const e16x8*>(xp + 16 * s);
  }
  __syncthreads();                                       // sblk visible
  if constexpr (LNK) {
    proj_block<PROJ_LN>(a, sblk[0], sblk, 0, nchunk, ring, xf, wave, lane, m0w, sbias);
  } else {
    for (int b = 0; b < a.nblk; ++b) {
      const ProjBlock pb = sblk[b];
      const int kind = __builtin_amdgcn_readfirstlane(pb.kind);
      // (bias copies alternate: the waves that are still in block b - 1's epilogue read the other one)
      if (kind == PROJ_VT) proj_block<PROJ_VT>(a, pb, sblk, b, nchunk, ring, xf, wave, lane, m0w, sbias + (b & 1) * 768);
      else if (kind == PROJ_KFRAG) proj_block<PROJ_KFRAG>(a, pb, sblk, b, nchunk, ring, xf, wave, lane, m0w, sbias + (b & 1) * 768);
      else if (kind == PROJ_VFRAG) proj_block<PROJ_VFRAG>(a, pb, sblk, b, nchunk, ring, xf, wave, lane, m0w, sbias + (b & 1) * 768);
      else proj_block<PROJ_ROW>(a, pb, sblk, b, nchunk, ring, xf, wave, lane, m0w, sbias + (b & 1) * 768);
    }
  }
}

int launch_proj256(const ProjArgs& a, hipStream_t st) {
  if (a.M <= 0 || !a.X || a.nblk < 1 || a.nblk > PROJ_MAX_BLOCKS || a.ldx % 8 || ((uintptr_t)a.X & 15)) ETD_FAIL(ETD_EINVAL, "proj256: bad arguments");
  for (int b = 0; b < a.nblk; ++b) {
    const ProjBlock& p = a.blk[b];
    if (!p.Wf || !p.bias || !p.dst || ((uintptr_t)p.Wf & 15) || ((uintptr_t)p.dst & 15)) ETD_FAIL(ETD_EINVAL, "proj256: bad block %d", b);
    if (p.kind == PROJ_VT && (a.S <= 0 || a.Spad < a.S)) ETD_FAIL(ETD_EINVAL, "proj256: bad V^T geometry");
    if ((p.kind == PROJ_ROW || p.kind == PROJ_LN) && (p.ldd % 8)) ETD_FAIL(ETD_EINVAL, "proj256: row stride must be a multiple of 8");
    if ((p.kind == PROJ_KFRAG || p.kind == PROJ_VFRAG) && (a.S <= 0 || a.S % 64 || a.kv_nstep != a.S / 64 || a.M % 32)) ETD_FAIL(ETD_EINVAL, "proj256: fragment images need sequences of a multiple of 64 tokens");
    if (p.kind == PROJ_LN && (!a.R || !a.gamma || !a.beta || a.nblk != 1)) ETD_FAIL(ETD_EINVAL, "proj256: an LN block needs residual + LayerNorm parameters and a launch of its own");
  }
  const char* pname = a.blk[0].kind == PROJ_LN ? "k_proj256_ln" : (a.nblk == 6 ? "k_proj256_kv6" : (a.nblk == 3 ? "k_proj256_qkv" : "k_proj256_row"));
  ETD_LAUNCH_FILTER(pname);
  ProfScope ps(pname, st, 2.0 * a.M * 256.0 * 256.0 * a.nblk, ((double)a.M * 256 * (1 + a.nblk) + 65536.0 * a.nblk) * 2);
  if (a.blk[0].kind == PROJ_LN) hipLaunchKernelGGL(k_proj256<true>, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_proj256<false>, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// [256 out][256 in] fp32 (nn.Linear layout) -> one block of k_proj256's stream: [chunk 4][k-step 4][tile 8][lane 64][8]
void pack_proj_weights(const float* W, bool permute_rows, uint16_t* dst, uint16_t (*f2bf)(float)) {
  for (int c = 0; c < 4; ++c)
    for (int kk = 0; kk < 4; ++kk)
      for (int t = 0; t < 8; ++t)
        for (int l = 0; l < 64; ++l) {
          const int r = l & 31, h = l >> 5, s = 4 * c + kk;
          int feat = 32 * t + r;
          if (permute_rows) { const int i = (r & 3) + 4 * (r >> 3), hh = (r >> 2) & 1; feat = 32 * t + 16 * (i >> 3) + 8 * hh + (i & 7); }
          uint16_t* d = dst + ((((size_t)c * 4 + kk) * 8 + t) * 64 + l) * 8;
          for (int j = 0; j < 8; ++j) d[j] = f2bf(W[(size_t)feat * 256 + 16 * s + 8 * h + j]);
        }
}

// ================================================================================================
// k_enc_layer: one whole EncoderLayer for one 256-token sequence (a frame's 256 frequency bins) per workgroup.
//     x = LN(x + fc_o(MHA(x)));  x = LN(x + fc_2(relu(fc_1(x))))          amt_apc.py:244-259, one shared LayerNorm
// 8 waves x 32 tokens.  Everything between the layer's input and output lives on the CU:
//   * per head: Q, K, V projections of the wave's own 32 tokens on the k_proj256 skeleton (X fragments from registers, weights
//     from the LDS ring).  Q stays in registers as the B operand of S^T = K Q^T; K leaves as MFMA A-fragments into an LDS image
//     [key tile][k-step][lane] (linear 1 KiB pieces: conflict-free writes and reads, no address arithmetic); V is produced
//     with the operands swapped (feature on the lane), whose accumulator registers 8 ks .. 8 ks + 7 ARE the A fragment of
//     O^T += V^T P^T for k-step ks in exactly the permuted key order in which P^T leaves the softmax (guide section 3) -- they go
//     to a second LDS image;
//   * flash-style softmax per 64 keys as in k_attn (query on the lane, statistics lane-local + one exchange with lane ^ 32);
//   * the normalised O^T accumulator is the B operand of the output projection (Wo's columns are packed in its permuted order);
//     fc_o of all four heads accumulates after the last head, then bias + residual + LayerNorm in registers, and the result --
//     laid out as B fragments -- feeds the feed-forward block of k_ffn_fused unchanged.
// Weight stream: 32 chunks of 32 KiB (12 QKV, 4 fc_o, 16 feed-forward) through the two-slot ring, 1 MiB per sequence from L2.
// HBM traffic per sequence: 128 KiB in, 128 KiB out (round 1: ~2.3 MiB).
// ================================================================================================
// The ring: five slots of 16 KiB; the 1 MiB stream is consumed as 64 half-chunks (16 fragments each: 16 MFMAs per wave), four
// of them requested ahead.  With two 32 KiB slots (one chunk ahead, ~1000 clocks of MFMAs) every chunk top waited for its DMA -- an
// L2 round trip under load is 2-4 k clocks and this kernel has ONE workgroup per CU, nobody else to fill the gap: SQ_WAIT_ANY was 50 %
// of the wave cycles (profiles/r02_pmc_extractor.txt).  Each wave issues 2 one-KiB pieces per half-chunk and waits with a COUNTED
// vmcnt: all but the 6 youngest (= the three half-chunks behind the one needed).  Other vector-memory operations between an issue and
// its wait (the X fragment reloads, compiler spills) only make that wait stricter, never laxer: vmcnt retires in order.
#define ENC_HC_ELEMS 8192
#define ENC_NSLOT 5
#define ENC_AHEAD 4
#define ENC_NHC 64
#define ENC_LDS_PAR 0
#define ENC_LDS_RING ((768 + 256 * 3 + 512 + 256) * 4)
#define ENC_LDS_K (ENC_LDS_RING + ENC_NSLOT * ENC_HC_ELEMS * 2)
#define ENC_LDS_V (ENC_LDS_K + 32768)
#define ENC_LDS_BYTES (ENC_LDS_V + 32768)

// LDS reads / writes as `opaque per-lane base register + immediate offset`.  This kernel's LDS is 155 KiB and a ds instruction's offset field reaches 64 KiB: handed
// plain pointers, hipcc folds every image's constant base (80 .. 147 KiB) into each access's constant part, finds it does not fit and keeps ONE address register per
// distinct access -- with the head loop unrolled that was ~100 live address registers, the source of the kernel's 91 spills and of accumulator copies between MFMAs
// (LABNOTES round 4).  The bases below pass through an empty asm, so each access is `ds_read_b128 v, base offset:imm`.
typedef __attribute__((address_space(3))) const e16x8* enc_lds_cp;
typedef __attribute__((address_space(3))) e16x8* enc_lds_p;
#define ENC_RD8(base, off) (*(enc_lds_cp)(uintptr_t)((base) + (unsigned)(off)))
#define ENC_WR8(base, off) (*(enc_lds_p)(uintptr_t)((base) + (unsigned)(off)))
__global__ __launch_bounds__(512, 2) void k_enc_layer(EncLayerArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ENC_LDS_BYTES];
  e16* ring = reinterpret_cast<e16*>(smem + ENC_LDS_RING);
  e16* Kimg = reinterpret_cast<e16*>(smem + ENC_LDS_K);        // [key tile 8][k-step 4][lane 64][8]
  e16* Vimg = reinterpret_cast<e16*>(smem + ENC_LDS_V);        // [key tile 8][ks 2][dt 2][lane 64][8]
  float* sbqkv = reinterpret_cast<float*>(smem + ENC_LDS_PAR);   // bqkv[768] | bo[256] | gamma[256] | beta[256] | b1[512] | b2[256]
  float* sbo = sbqkv + 768; float* sg = sbo + 256; float* sbe = sg + 256; float* sb1 = sbe + 256; float* sb2 = sb1 + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const long long row = (long long)blockIdx.x * 256 + wave * 32 + r;        // this lane's token

  // byte addresses of this lane's 16 bytes in: ring slot 0 (slots 0 .. 3 by immediate) / ring slot 4; the K / V images (reads: tile by immediate; writes: this wave's block)
  unsigned ringA = (unsigned)reinterpret_cast<uintptr_t>(ring) + lane * 16, ringB = ringA + 4 * ENC_HC_ELEMS * 2;
  unsigned kimg_l = (unsigned)reinterpret_cast<uintptr_t>(Kimg) + lane * 16, vimg_l = (unsigned)reinterpret_cast<uintptr_t>(Vimg) + lane * 16;
  unsigned kimg_w = kimg_l + wave * 4096, vimg_w = vimg_l + wave * 4096;
  asm volatile("" : "+v"(ringA), "+v"(ringB), "+v"(kimg_l), "+v"(vimg_l), "+v"(kimg_w), "+v"(vimg_w));
#define ENC_SLOT(hc) (((hc) % ENC_NSLOT) == 4 ? ringB : ringA + ((hc) % ENC_NSLOT) * (ENC_HC_ELEMS * 2))
  auto issue = [&](int hc) {                             // half-chunk hc -> slot hc % 5: 16 one-KiB pieces, 2 per wave
    const e16* src = a.Wl + (long long)hc * ENC_HC_ELEMS + wave * (2 * 512) + lane * 8;
    e16* dst = ring + (hc % ENC_NSLOT) * ENC_HC_ELEMS + wave * (2 * 512);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  };
  // top of a half-chunk: it has landed for this wave (counted wait) and, behind the barrier, for everybody; the slot of half-chunk
  // hc - 1 is free (every wave has passed this barrier, so it is done reading it) and takes half-chunk hc + 4
#ifdef ETD_ENC_SYNCTHREADS      /* A/B build: the round-2 form */
#define ENC_BARRIER() __syncthreads()
#else
#define ENC_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#define ENC_TOP(hc)                                                                                                    \
  do {                                                                                                                 \
    if ((hc) + ENC_AHEAD - 1 < ENC_NHC) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                  \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                   \
    ENC_BARRIER();                    /* NOT __syncthreads(): its fence waits vmcnt(0) while LDS-DMA is pending and drains the ring */ \
    if ((hc) + ENC_AHEAD < ENC_NHC) issue((hc) + ENC_AHEAD);                                                           \
  } while (0)
  // 16 fragments of a half-chunk against B (or A) operands, four at a time, one group requested ahead of the MFMAs that use it
#define ENC_HALF(sl, BODY)                                                                                             \
  {                                                                                                                    \
    e16x8 af[2][4];                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) af[0][k] = ENC_RD8(sl, k * 1024);                                   \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                                 \
      if (gq < 3) { _Pragma("unroll") for (int k = 0; k < 4; ++k) af[(gq + 1) & 1][k] = ENC_RD8(sl, ((gq + 1) * 4 + k) * 1024); }   \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) { const e16x8 fa = af[gq & 1][k]; BODY }                          \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
  }

  // the same with TWO fragments per group (fc_o and the feed-forward block: 128 accumulator registers + the 64-register token tile leave no room for two
  // 4-fragment buffers -- with them hipcc parked 91 registers of the LayerNorm'ed tile in scratch once per sequence; the partner wave of the SIMD hides the shorter lead)
#define ENC_HALF2(sl, BODY)                                                                                            \
  {                                                                                                                    \
    e16x8 af[2][2];                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) af[0][k] = ENC_RD8(sl, k * 1024);                                   \
    _Pragma("unroll") for (int g2 = 0; g2 < 8; ++g2) {                                                                 \
      if (g2 < 7) { _Pragma("unroll") for (int k = 0; k < 2; ++k) af[(g2 + 1) & 1][k] = ENC_RD8(sl, ((g2 + 1) * 2 + k) * 1024); }   \
      _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) { const e16x8 fa = af[g2 & 1][k2]; const int gq = g2 >> 1, k = 2 * (g2 & 1) + k2; (void)gq; (void)k; BODY }   \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
  }

#pragma unroll
  for (int i = 0; i < ENC_AHEAD; ++i) issue(i);
  for (int i = tid; i < 768; i += 512) sbqkv[i] = a.bqkv[i];
  if (tid < 256) { sbo[tid] = a.bo[tid]; sg[tid] = a.gamma[tid]; sbe[tid] = a.beta[tid]; sb2[tid] = a.b2[tid]; }
  sb1[tid] = a.b1[tid];

  const float qscale = 0.125f * 1.4426950408889634f;     // 1 / sqrt(64) and the base-2 exponent, folded into Q
  e16x8 ofr[16];                                         // the four heads' normalised outputs as B fragments (k-step 4 head + ks)
  // ---- the wave's 32 tokens as B / A fragments, loaded ONCE and held through the four heads and the residual (rounds 2-3 re-read them per head -- 80 fragment-shaped
  // loads per wave, 32 rows x 32 bytes per instruction -- to save 64 registers during the attention; since the LDS accesses stopped costing ~100 address registers
  // (ENC_RD8 / ENC_WR8) there is room)
  e16x8 xf[16];
  {
    const e16* xp = a.X + row * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const e16x8*>(xp + 16 * s);
  }
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) {
    e16x8 qf[4];
    // ---- Q (part 0), K (part 1): token on the lane; V (part 2): feature on the lane.  A part = two half-chunks (k-steps 0..7, 8..15)
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      f32x16 acc[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;
#pragma unroll
      for (int hf2 = 0; hf2 < 2; ++hf2) {
        const int hc = (hd * 3 + part) * 2 + hf2;
        ENC_TOP(hc);
        const unsigned sl = ENC_SLOT(hc);
        // fragment 4 gq + k of this half = k-step 8 hf2 + 2 gq + (k >> 1), tile k & 1
        if (part == 2) ENC_HALF(sl, acc[k & 1] = mfma32(xf[8 * hf2 + 2 * gq + (k >> 1)], fa, acc[k & 1]);)
        else           ENC_HALF(sl, acc[k & 1] = mfma32(fa, xf[8 * hf2 + 2 * gq + (k >> 1)], acc[k & 1]);)
      }
      if (part < 2) {
        // accumulator (tile dt, register i) of lane half h = feature 32 dt + 16 (i >> 3) + 8 h + (i & 7) of this head's 64:
        // fragment s = 2 dt + (i >> 3), element i & 7
        e16x8 fr[4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 bb = *reinterpret_cast<const f32x4*>(sbqkv + part * 256 + hd * 64 + 32 * dt + 16 * u + 8 * h + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float v = acc[dt][8 * u + 4 * q + j] + bb[j];
                if (part == 0) v *= qscale;
                fr[2 * dt + u][4 * q + j] = (e16)v;
              }
            }
        if (part == 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s) qf[s] = fr[s];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) ENC_WR8(kimg_w, s * 1024) = fr[s];
        }
      } else {
        // acc[dt][i]: key (i & 3) + 8 (i >> 2) + 4 h of the wave's 32, feature 32 dt + r; registers 8 ks .. + 8 = A fragment of k-step ks
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float bv = sbqkv[512 + hd * 64 + 32 * dt + r];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            e16x8 fr;
#pragma unroll
            for (int j = 0; j < 8; ++j) fr[j] = (e16)(acc[dt][8 * ks + j] + bv);
            ENC_WR8(vimg_w, (ks * 2 + dt) * 1024) = fr;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // K and V images of this head complete (raw: the ring's DMAs stay in flight)
    // ---- attention of the wave's 32 queries against the 256 keys, 64 keys per step
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      f32x16 sT[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sT[kt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          sT[kt] = mfma32(ENC_RD8(kimg_l, ((2 * kp + kt) * 4 + s) * 1024), qf[s], sT[kt]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sT[kt][i]);
      mx = fmaxf(mx, xhalf(mx));
      const float mnew = fmaxf(mrun, mx);
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
      mrun = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float p = __builtin_amdgcn_exp2f(sT[kt][i] - mnew); sT[kt][i] = p; ps += p; }
      lrun = lrun * alpha + ps;
      // (skipping this rescale when no query of the wave raised its maximum -- a wave-uniform branch -- measured nothing: 0.630 vs 0.632 ms)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          e16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (e16)sT[kt][8 * ks + j];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            o[dt] = mfma32(ENC_RD8(vimg_l, (((2 * kp + kt) * 2 + ks) * 2 + dt) * 1024), pf, o[dt]);
        }
    }
    lrun += xhalf(lrun);
    const float inv = 1.f / lrun;
    // O^T[d][query]: registers 8 u .. + 8 of tile dt = B fragment of k-step 2 dt + u (d = 32 dt + 16 u + 8 (j >> 2) + 4 h + (j & 3))
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) ofr[4 * hd + 2 * dt + u][j] = (e16)(o[dt][8 * u + j] * inv);
  }

  // ---- fc_o over the four heads (half-chunks 24 .. 31: head hd = k-steps 4 hd .. + 4, two per half), bias + residual + LayerNorm
  f32x16 acc2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
#pragma unroll
  for (int hd = 0; hd < 4; ++hd)
#pragma unroll
    for (int hf2 = 0; hf2 < 2; ++hf2) {
      const int hc = 24 + hd * 2 + hf2;
      ENC_TOP(hc);
      const unsigned sl = ENC_SLOT(hc);
      // fragment 4 gq + k of this half = k-step 2 hf2 + (gq >> 1) of the head, tile 4 (gq & 1) + k
      ENC_HALF2(sl, acc2[4 * (gq & 1) + k] = mfma32(fa, ofr[4 * hd + 2 * hf2 + (gq >> 1)], acc2[4 * (gq & 1) + k]);)
    }
  // LayerNorm of (acc2 + bias + resid) -> xf (e16 fragments); statistics in fp32.
  // (The parameter vectors sit at the BOTTOM of this kernel's LDS: a ds_read reaches 64 KiB from its base register with its offset field.  In rounds 2-3 they sat
  // above the ring and the K / V images, at 0x24000: hipcc then materialised all 96 read addresses (0x25000 | lane part ...) in registers, kept them for the second
  // LayerNorm behind the feed-forward block and spilled them there -- the kernel's 91 spilled registers were addresses.)
  const float* lnp_bo = sbo; const float* lnp_b2 = sb2; const float* lnp_g = sg; const float* lnp_be = sbe;
  int lnoff = 8 * h;                    // (opaque to the scheduler from tile to tile, see the macro)
#define ENC_RESID_LN(BIAS)                                                                                                   \
  {                                                                                                                          \
    float s1 = 0.f;                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                          \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                          \
        const f32x4 bb = *reinterpret_cast<const f32x4*>((BIAS) + lnoff + 32 * t + 16 * u + 4 * q);                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
          const float v = acc2[t][8 * u + 4 * q + j] + bb[j] + bf2f(xf[2 * t + u][4 * q + j]);                              \
          acc2[t][8 * u + 4 * q + j] = v; s1 += v;                                                                           \
        }                                                                                                                    \
      }                                                                                                                      \
      asm volatile("" : "+v"(lnoff) : "v"(s1));   /* (tile t + 1's parameter reads depend on tile t's sum: the scheduler otherwise requests all 96 parameter vectors up front and the allocator spills them) */ \
    }                                                                                                                        \
    s1 += xhalf(s1);                                                                                                         \
    const float mean = s1 * (1.f / 256.f);                                                                                   \
    float s2 = 0.f;                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) _Pragma("unroll") for (int i = 0; i < 16; ++i) { const float d = acc2[t][i] - mean; s2 += d * d; }   \
    s2 += xhalf(s2);                                                                                                         \
    const float rstd = rsqrtf(s2 * (1.f / 256.f) + 1e-5f);                                                                   \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                          \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                          \
        const int f0 = lnoff + 32 * t + 16 * u + 4 * q;                                                                      \
        const f32x4 gg = *reinterpret_cast<const f32x4*>(lnp_g + f0), be = *reinterpret_cast<const f32x4*>(lnp_be + f0);    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) xf[2 * t + u][4 * q + j] = (e16)((acc2[t][8 * u + 4 * q + j] - mean) * rstd * gg[j] + be[j]);   \
      }                                                                                                                      \
      asm volatile("" : "+v"(lnoff) : "v"(xf[2 * t + 1]));                                                                   \
    }                                                                                                                        \
  }
  ENC_RESID_LN(lnp_bo)

  // ---- feed-forward block (half-chunks 32 .. 63: per 32 hidden features one half of W1 fragments, one of W2 fragments)
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
  for (int sc = 0; sc < FFN_NSUB; ++sc) {
    const int hc = 32 + 2 * sc;
    ENC_TOP(hc);
    const unsigned sl = ENC_SLOT(hc);
    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
    ENC_HALF2(sl, acc1 = mfma32(fa, xf[gq * 4 + k], acc1);)
    e16x8 hfr[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + 32 * sc + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) hfr[q >> 1][4 * (q & 1) + j] = (e16)fmaxf(acc1[4 * q + j] + bb[j], 0.f);
    }
    ENC_TOP(hc + 1);
    const unsigned sl2 = ENC_SLOT(hc + 1);
    ENC_HALF2(sl2, acc2[4 * (gq & 1) + k] = mfma32(fa, hfr[gq >> 1], acc2[4 * (gq & 1) + k]);)
  }
  ENC_RESID_LN(lnp_b2)
  e16* yp = a.Y + row * 256 + 8 * h;
#pragma unroll
  for (int s = 0; s < 16; ++s) *reinterpret_cast<e16x8*>(yp + 16 * s) = xf[s];
#undef ENC_RESID_LN
#undef ENC_HALF
#undef ENC_HALF2
#undef ENC_SLOT
#undef ENC_TOP
}

int launch_enc_layer(const EncLayerArgs& a, hipStream_t st) {
  if (a.n_seq <= 0 || !a.X || !a.Wl || !a.bqkv || !a.bo || !a.gamma || !a.beta || !a.b1 || !a.b2 || !a.Y || (((uintptr_t)a.X | (uintptr_t)a.Y | (uintptr_t)a.Wl) & 15))
    ETD_FAIL(ETD_EINVAL, "enc_layer: bad arguments");
  const double tok = (double)a.n_seq * 256;
  ETD_LAUNCH_FILTER("k_enc_layer");
  ProfScope ps("k_enc_layer", st, 2.0 * tok * 256 * (768 + 256 + 1024) + 4.0 * a.n_seq * 256.0 * 256 * 256, tok * 256 * 2 * 2 + 1048576.0);
  hipLaunchKernelGGL(k_enc_layer, dim3(a.n_seq), dim3(512), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

void pack_enc_layer_weights(const float* Wq, const float* Wk, const float* Wv, const float* Wo, const float* W1, const float* W2, uint16_t* dst, uint16_t (*f2bf)(float)) {
  auto perm = [](int r) { const int i = (r & 3) + 4 * (r >> 3), hh = (r >> 2) & 1; return 16 * (i >> 3) + 8 * hh + (i & 7); };   // A-row r -> feature offset in its 32-tile
  for (int hd = 0; hd < 4; ++hd)
    for (int part = 0; part < 3; ++part) {
      const float* W = part == 0 ? Wq : (part == 1 ? Wk : Wv);
      uint16_t* ch = dst + (size_t)(hd * 3 + part) * FFN_SLOT_ELEMS;
      for (int s = 0; s < 16; ++s)
        for (int dt = 0; dt < 2; ++dt)
          for (int l = 0; l < 64; ++l) {
            const int r = l & 31, h = l >> 5;
            const int rowf = 64 * hd + 32 * dt + (part == 2 ? r : perm(r));
            uint16_t* d = ch + ((size_t)(s * 2 + dt) * 64 + l) * 8;
            for (int j = 0; j < 8; ++j) d[j] = f2bf(W[(size_t)rowf * 256 + 16 * s + 8 * h + j]);
          }
    }
  for (int hd = 0; hd < 4; ++hd) {
    uint16_t* ch = dst + (size_t)(12 + hd) * FFN_SLOT_ELEMS;
    for (int ks = 0; ks < 4; ++ks)
      for (int t = 0; t < 8; ++t)
        for (int l = 0; l < 64; ++l) {
          const int r = l & 31, h = l >> 5;
          const int rowf = 32 * t + perm(r);
          uint16_t* d = ch + ((size_t)(ks * 8 + t) * 64 + l) * 8;
          for (int j = 0; j < 8; ++j) d[j] = f2bf(Wo[(size_t)rowf * 256 + 64 * hd + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)]);
        }
  }
  pack_ffn_weights(W1, W2, dst + (size_t)16 * FFN_SLOT_ELEMS, f2bf);
}


#ifdef ETD_EXPERIMENTS      // k_post_attn: a measured dead end (equal in time to the two launches it replaces, 103 spilled registers), kept for the record
// ================================================================================================
// k_post_attn: everything of a layer that follows its attention, for layers whose attention runs as its own launch (the
// frequency decoder's self- and cross-attention blocks, the time decoder):
//     x1 = LN(resid + AO Wo^T + bo);   [y = LN(x1 + relu(x1 W1^T + b1) W2^T + b2)]      amt_apc.py:250-259, 281-286, 306-320
// = k_proj256's LayerNorm block whose output fragments feed k_ffn_fused's loop without leaving the registers (the tail of
// k_enc_layer with the attention output read from memory).  4 waves x 32 tokens, two workgroups per CU.
// ================================================================================================
__global__ __launch_bounds__(256, 2) void k_post_attn(PostAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FFN_SLOT_ELEMS * 2 + (256 * 3 + 512 + 256) * 4];
  e16* ring = reinterpret_cast<e16*>(smem);
  float* sbo = reinterpret_cast<float*>(smem + 2 * FFN_SLOT_ELEMS * 2);      // bo[256] | gamma[256] | beta[256] | b1[512] | b2[256]
  float* sg = sbo + 256; float* sbe = sg + 256; float* sb1 = sbe + 256; float* sb2 = sb1 + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;
  const int nstep = a.Wffn ? 20 : 4;
  auto issue = [&](int st) {                             // steps 0..3: fc_o chunks, 4..19: feed-forward chunks; 8 pieces per wave
    const e16* base = st < 4 ? a.Wo + (long long)st * FFN_SLOT_ELEMS : a.Wffn + (long long)(st - 4) * FFN_SLOT_ELEMS;
    const e16* src = base + wave * (8 * 512) + lane * 8;
    e16* dst = ring + (st & 1) * FFN_SLOT_ELEMS + wave * (8 * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  };
#define PA_TOP(g) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if ((g) + 1 < nstep) issue((g) + 1); } while (0)
  issue(0);
  e16x8 xf[16];
  {
    const e16* xp = a.AO + (long long)mc * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const e16x8*>(xp + 16 * s);
  }
  sbo[tid] = a.bo[tid]; sg[tid] = a.gamma[tid]; sbe[tid] = a.beta[tid];
  if (a.Wffn) { sb1[tid] = a.b1[tid]; sb1[256 + tid] = a.b1[256 + tid]; sb2[tid] = a.b2[tid]; }
  f32x16 acc2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    PA_TOP(c);
    const e16* sl = ring + (c & 1) * FFN_SLOT_ELEMS + lane * 8;
    e16x8 af[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + k * 512);
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {                     // group gq: k-step 4 c + (gq >> 1), tiles 4 (gq & 1) .. + 4
      if (gq < 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) af[(gq + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + ((gq + 1) * 4 + k) * 512);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc2[4 * (gq & 1) + k] = mfma32(af[gq & 1][k], xf[4 * c + (gq >> 1)], acc2[4 * (gq & 1) + k]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    const int rrow = a.r_mod > 0 ? mc % a.r_mod : mc;
    const e16* rp = a.R + (long long)rrow * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const e16x8*>(rp + 16 * s);       // residual
  }
#define PA_RESID_LN(BIAS)                                                                                                    \
  {                                                                                                                          \
    float s1 = 0.f;                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int q = 0; q < 2; ++q) {   \
      const f32x4 bb = *reinterpret_cast<const f32x4*>((BIAS) + 32 * t + 16 * u + 8 * h + 4 * q);                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
        const float v = acc2[t][8 * u + 4 * q + j] + bb[j] + bf2f(xf[2 * t + u][4 * q + j]);                                \
        acc2[t][8 * u + 4 * q + j] = v; s1 += v;                                                                             \
      }                                                                                                                      \
    }                                                                                                                        \
    s1 += xhalf(s1);                                                                                                         \
    const float mean = s1 * (1.f / 256.f);                                                                                   \
    float s2 = 0.f;                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) _Pragma("unroll") for (int i = 0; i < 16; ++i) { const float d = acc2[t][i] - mean; s2 += d * d; }   \
    s2 += xhalf(s2);                                                                                                         \
    const float rstd = rsqrtf(s2 * (1.f / 256.f) + 1e-5f);                                                                   \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int q = 0; q < 2; ++q) {   \
      const int f0 = 32 * t + 16 * u + 8 * h + 4 * q;                                                                        \
      const f32x4 gg = *reinterpret_cast<const f32x4*>(sg + f0), be = *reinterpret_cast<const f32x4*>(sbe + f0);            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) xf[2 * t + u][4 * q + j] = (e16)((acc2[t][8 * u + 4 * q + j] - mean) * rstd * gg[j] + be[j]);   \
    }                                                                                                                        \
  }
  PA_RESID_LN(sbo)
  if (a.Wffn) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
    for (int sc = 0; sc < FFN_NSUB; ++sc) {
      const int g = 4 + sc;
      PA_TOP(g);
      const e16* sl = ring + (g & 1) * FFN_SLOT_ELEMS + lane * 8;
      f32x16 acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
      e16x8 af[2][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + k * 512);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (gq < 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) af[(gq + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + ((gq + 1) * 4 + k) * 512);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) af[0][k] = *reinterpret_cast<const e16x8*>(sl + (16 + k) * 512);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc1 = mfma32(af[gq & 1][k], xf[gq * 4 + k], acc1);
        __builtin_amdgcn_sched_barrier(0);
      }
      e16x8 hf[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + 32 * sc + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) hf[q >> 1][4 * (q & 1) + j] = (e16)fmaxf(acc1[4 * q + j] + bb[j], 0.f);
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (gq < 3) {
#pragma unroll
          for (int k = 0; k < 4; ++k) af[(gq + 1) & 1][k] = *reinterpret_cast<const e16x8*>(sl + (16 + (gq + 1) * 4 + k) * 512);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc2[4 * (gq & 1) + k] = mfma32(af[gq & 1][k], hf[gq >> 1], acc2[4 * (gq & 1) + k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    PA_RESID_LN(sb2)
  }
  if (m < a.M) {
    e16* yp = a.Y + (long long)m * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) *reinterpret_cast<e16x8*>(yp + 16 * s) = xf[s];
  }
#undef PA_RESID_LN
#undef PA_TOP
}

int launch_post_attn(const PostAttnArgs& a, hipStream_t st) {
  if (a.M <= 0 || !a.AO || !a.R || !a.Wo || !a.bo || !a.gamma || !a.beta || !a.Y || (a.Wffn && (!a.b1 || !a.b2)) ||
      (((uintptr_t)a.AO | (uintptr_t)a.R | (uintptr_t)a.Y | (uintptr_t)a.Wo | (uintptr_t)a.Wffn) & 15))
    ETD_FAIL(ETD_EINVAL, "post_attn: bad arguments");
  ETD_LAUNCH_FILTER(a.Wffn ? "k_post_attn_ffn" : "k_post_attn");
  ProfScope ps(a.Wffn ? "k_post_attn_ffn" : "k_post_attn", st, 2.0 * a.M * 256.0 * (256.0 + (a.Wffn ? 1024.0 : 0.0)), (double)a.M * 256 * 2 * 3 + 655360.0);
  hipLaunchKernelGGL(k_post_attn, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
#else
int launch_post_attn(const PostAttnArgs&, hipStream_t) { ETD_FAIL(ETD_EINVAL, "k_post_attn is built with -DETD_EXPERIMENTS only"); }
#endif      // ETD_EXPERIMENTS

// ================================================================================================
// k_attn_frag: softmax(Q K^T / 8) V per (sequence, head) with K and V arriving as MFMA-fragment images (k_proj256's KFRAG /
// VFRAG blocks): a 64-key step is one contiguous 16 KiB piece of global memory that goes into an LDS ring by LDS-DMA and is
// read back as linear, conflict-free ds_read_b128 fragments -- no register staging, no transposed V copy, no address
// arithmetic in the loop.  The arithmetic is k_attn's (query on the lane, online softmax in fp32, P^T from the accumulator
// registers).  Workgroup = 4 waves x 32 queries; three ring slots, two steps in flight.            amt_apc.py:349-368
// ================================================================================================
#define AF_SLOT_ELEMS 8192              // e16 elements per ring slot (16 KiB)
__global__ __launch_bounds__(256) void k_attn_frag(AttnFragArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * AF_SLOT_ELEMS * 2];
  e16* ring = reinterpret_cast<e16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int seq = blockIdx.y >> 2, head = blockIdx.y & 3;
  const int q0 = blockIdx.x * 128 + wave * 32;
  int qi = q0 + r; const bool qvalid = qi < a.Sq; if (!qvalid) qi = a.Sq - 1;
  const int nstep = a.Sk >> 6;
  const e16* img = a.KV + (long long)(seq * 4 + head) * nstep * AF_SLOT_ELEMS;
  auto issue = [&](int st) {                             // step st -> slot st % 3: 16 one-KiB pieces, 4 per wave
    const e16* src = img + (long long)st * AF_SLOT_ELEMS + wave * (4 * 512) + lane * 8;
    e16* dst = ring + (st % 3) * AF_SLOT_ELEMS + wave * (4 * 512);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  };
  const e16* qp = a.Q + seq * a.q_seq_stride + (long long)qi * a.ldq + head * 64;
  e16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const e16x8*>(qp + s * 16 + h * 8);
  issue(0);
  if (nstep > 1) issue(1);
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  const float c = a.scale_log2e;
  for (int st = 0; st < nstep; ++st) {
    // the pieces of step st have landed (the 4 of step st + 1, issued after them, may still be in flight), for every wave
    if (st + 1 < nstep) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 2 < nstep) issue(st + 2);                   // into the slot of step st - 1, which every wave has left (it passed this barrier)
    const e16* sl = ring + (st % 3) * AF_SLOT_ELEMS + lane * 8;
    f32x16 sT[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sT[kt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) sT[kt] = mfma32(*reinterpret_cast<const e16x8*>(sl + (kt * 4 + s) * 512), qf[s], sT[kt]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sT[kt][i]);
    mx = fmaxf(mx, xhalf(mx));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
    mrun = mnew;
    const float mc = -mnew * c;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float p = __builtin_amdgcn_exp2f(fmaf(sT[kt][i], c, mc)); sT[kt][i] = p; ps += p; }
    lrun = lrun * alpha + ps;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        e16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (e16)sT[kt][8 * ks + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o[dt] = mfma32(*reinterpret_cast<const e16x8*>(sl + 4096 + ((kt * 2 + ks) * 2 + dt) * 512), pf, o[dt]);
      }
  }
  lrun += xhalf(lrun);
  const float inv = 1.f / lrun;
  if (qvalid) {
    e16* op = a.O + seq * a.o_seq_stride + (long long)(q0 + r) * a.ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + 8 * q + 4 * h;
        *reinterpret_cast<e16x4*>(op + d) = pack4e(o[dt][4 * q] * inv, o[dt][4 * q + 1] * inv, o[dt][4 * q + 2] * inv, o[dt][4 * q + 3] * inv);
      }
  }
}

int launch_attn_frag(const AttnFragArgs& a, hipStream_t st) {
  if (a.Sq <= 0 || a.Sk <= 0 || a.Sk % 64 || a.n_seq <= 0 || !a.Q || !a.KV || !a.O || (((uintptr_t)a.Q | (uintptr_t)a.KV) & 15) || a.ldq % 8 || a.ldo % 4)
    ETD_FAIL(ETD_EINVAL, "attn_frag: bad arguments (Sk must be a multiple of 64)");
  ETD_LAUNCH_FILTER("k_attn_frag");
  ProfScope ps("k_attn_frag", st, 1024.0 * a.n_seq * a.Sq * a.Sk, ((double)a.n_seq * (2.0 * a.Sq + 2.0 * a.Sk) * 256) * 2);
  hipLaunchKernelGGL(k_attn_frag, dim3((a.Sq + 127) / 128, a.n_seq * 4), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
