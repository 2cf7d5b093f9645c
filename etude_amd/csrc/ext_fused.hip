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

#define FFN_SLOT_ELEMS (16 * 1024)      // bf16 elements per ring slot: 32 fragments of 512 elements (1 KiB)
#define FFN_NSUB 16                     // 512 hidden features / 32

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256, 2) void k_ffn_fused(FfnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FFN_SLOT_ELEMS * 2 + (512 + 3 * 256) * 4];
  bf16* ring = reinterpret_cast<bf16*>(smem);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * FFN_SLOT_ELEMS * 2);      // b1[512] | b2[256] | gamma[256] | beta[256]
  float* sb2 = sb1 + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128, m = m0 + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;

  // ---- ring slot `sl` <- the 32 KiB of sub-chunk `sc`: 32 one-KiB pieces, 8 per wave
  auto issue = [&](int sc, int sl) {
    const bf16* src = a.Wf + (long long)sc * FFN_SLOT_ELEMS + wave * (8 * 512) + lane * 8;
    bf16* dst = ring + sl * FFN_SLOT_ELEMS + wave * (8 * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 512), (lptr_t)(dst + i * 512), 16, 0, 0);
  };
  issue(0, 0);
  // the token tile as B fragments: lane (token r, half h) holds X[token][16 s + 8 h .. + 8] for s = 0 .. 15
  bf16x8 xf[16];
  {
    const bf16* xp = a.X + (long long)mc * 256 + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xp + 16 * s);
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
    const bf16* sl = ring + (sc & 1) * FFN_SLOT_ELEMS + lane * 8;
    // hidden features 32 sc .. + 32 of the wave's 32 tokens
    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc1 = mfma32(*reinterpret_cast<const bf16x8*>(sl + s * 512), xf[s], acc1);
    bf16x8 hf[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + 32 * sc + 8 * q + 4 * h);       // register 4 q + j is hidden feature 32 sc + 8 q + 4 h + j
#pragma unroll
      for (int j = 0; j < 4; ++j) hf[q >> 1][4 * (q & 1) + j] = (bf16)fmaxf(acc1[4 * q + j] + bb[j], 0.f);
    }
    // out[256] += W2[:, these 32 hidden features] . hidden   (two 16-deep k-steps, 8 output tiles)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc2[t] = mfma32(*reinterpret_cast<const bf16x8*>(sl + (16 + ks * 8 + t) * 512), hf[ks], acc2[t]);
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
  bf16* yp = a.Y + (long long)m * 256 + 8 * h;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f0 = 32 * t + 16 * u + 8 * h + 4 * q;
        const f32x4 gg = *reinterpret_cast<const f32x4*>(sb2 + 256 + f0), be = *reinterpret_cast<const f32x4*>(sb2 + 512 + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * q + j] = (bf16)((acc2[t][8 * u + 4 * q + j] - mean) * rstd * gg[j] + be[j]);
      }
      if (m < a.M) *reinterpret_cast<bf16x8*>(yp + 16 * (2 * t + u)) = o;
    }
}

int launch_ffn_fused(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0 || !a.X || !a.Wf || !a.b1 || !a.b2 || !a.gamma || !a.beta || !a.Y || (((uintptr_t)a.X | (uintptr_t)a.Y | (uintptr_t)a.Wf) & 15))
    ETD_FAIL(ETD_EINVAL, "ffn_fused: bad arguments");
  ProfScope ps("k_ffn_fused", st, 2.0 * a.M * 256.0 * 512.0 * 2.0, (double)a.M * 256 * 2 * 2 + 512.0 * 256 * 2 * 2);
  hipLaunchKernelGGL(k_ffn_fused, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// Host side: fc_1 [512][256] and fc_2 [256][512] (fp32, nn.Linear layout) -> the kernel's weight stream, bf16:
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
