// Batched prefill of the EtudeDecoder, bf16: the MLP branch, attention.dense, the parallel residual and the NEXT layer's two
// LayerNorms of a GPT-NeoX layer in ONE launch                      modeling_gpt_neox.py:239-245 (mlp), :250-272 (layer)
//
//   h_out = h_in + [W2 gelu(W1 x2 + b1) + Wd attn + (b2 + bd)],   x1' = LN1'(h_out),  x2' = LN2'(h_out)
//
// The default path is three launches per layer (up + GELU -> Xcat, (down | dense) + residual, LayerNorm rows) that move 18 KB per
// token through HBM and run their 128 x 256 tiles at 460-520 TFLOP/s -- bound by L2 -> CU traffic (52 FLOP per byte the tile pulls
// from L2).  Here the token tile stays in registers for the whole branch, as in the extractor's k_ffn_fused.
//
// STATUS: correct and bit-identical, but OPT-IN (ETD_FUSED_PMLP=1) -- measured 290 us per launch (54 prompts x ~340 tokens)
// against 166 us for the two GEMMs + 16 us for the row kernel it replaces (tools/runs/r2_run47.sh: the job 574 -> 558 audio-s/s).
// A token's input fragments (128 registers) and its 512 fp32 outputs (256) leave 128 of a wave's 512 registers for everything
// else: hipcc parks part of the fragments in AGPRs (272 v_accvgpr moves per 64-MFMA chunk) and keeps only two A-fragment buffers
// in the first GEMM, so with ONE wave per SIMD every MFMA waits out the LDS read issued one MFMA earlier (~24 % MFMA duty).
// What it would take: the x2 fragments of half the K range in LDS (64 registers back), or MFMAs in inline asm with the B operand
// read from AGPRs directly.
//   * a wave owns 32 tokens; x2 enters once as the 32 B-operand fragments of v_mfma_f32_32x32x16_bf16 (128 registers);
//   * the 2048-wide hidden layer exists 32 features at a time: acc1 = W1[32 rows] . x2 (32 chained MFMAs), bias + erf-GELU +
//     bf16 rounding in registers; two v_permlane32_swap per k-step turn the accumulator's row order into the natural k order of a
//     B fragment, so the second GEMM multiplies exactly the operands the unfused (down | dense) GEMM would have read from Xcat,
//     in the same order: the result is BIT-IDENTICAL to the three-launch path (tests/test_gpu_decoder.py);
//   * the token's 512 outputs are 16 accumulator tiles = 256 registers (the AGPR half of the wave's 512): one wave per SIMD,
//     4 waves = 128 tokens per workgroup, one workgroup per CU;
//   * W2's / Wd's output rows are permuted on the host so that accumulator register (tile t, i) of lane half h is feature
//     32 t + 16 (i >> 3) + 8 h + (i & 7): 8 consecutive features per lane -- the residual row is read and written in 32-byte
//     pieces and the group sums of the LayerNorm are the very partial sums k_ln_rows forms per lane, folded in its order;
//   * weights: one stream per layer in fragment order (1 KiB per fragment, lane l's 16 bytes at 16 l), 64 KiB chunks
//     [down(k - 1) | up(k)] so that the GELU of chunk k - 1 (VALU) runs beside the MFMAs of up(k); two ring slots filled by
//     LDS-DMA (global_load_lds), one barrier per chunk; 256 FLOP per weight byte pulled from L2.
#include "dec_epilogue.h"
#include "dec_kernels.h"
#include "prof.h"

#define PM_SLOT_ELEMS (32 * 1024)          // bf16 elements per ring slot: 64 fragments of 512 elements (64 KiB)

typedef const __attribute__((address_space(1))) void* pm_gptr_t;
typedef __attribute__((address_space(3))) void* pm_lptr_t;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dmlp_fused(DMlpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PM_SLOT_ELEMS * 2 + 2048 * 4];
  bf16* ring = reinterpret_cast<bf16*>(smem);
  float* sbu = reinterpret_cast<float*>(smem + 2 * PM_SLOT_ELEMS * 2);      // b_up[2048]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;

  // ring slot (k & 1) <- stream chunk k: 64 one-KiB pieces, 16 per wave
  auto issue = [&](int k) {
    const bf16* src = a.Wm + (long long)k * PM_SLOT_ELEMS + wave * (16 * 512) + lane * 8;
    bf16* dst = ring + (k & 1) * PM_SLOT_ELEMS + wave * (16 * 512);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_global_load_lds((pm_gptr_t)(src + i * 512), (pm_lptr_t)(dst + i * 512), 16, 0, 0);
  };
#define PM_TOP(k)                                                                                                     \
  do {                                                                                                                \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* this wave's pieces of chunk k have landed ... */              \
    __syncthreads();                                    /* ... and everybody's; every wave is done reading the other slot */ \
    if ((k) + 1 < DMLP_NCHUNK) issue((k) + 1);                                                                          \
  } while (0)

  issue(0);
  // the token tile as B fragments: lane (token r, half h) holds x2[token][16 s + 8 h .. + 8], s = 0 .. 31
  bf16x8 xf[32];
  {
    const bf16* xp = a.X2 + (long long)mc * 512 + 8 * h;
#pragma unroll
    for (int s = 0; s < 32; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xp + 16 * s);
  }
  for (int i = tid; i < 2048; i += 256) sbu[i] = a.b_up[i];

  f32x16 acc2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
  f32x16 acc1;

  // ---- up(k): acc = W1[32 k .. + 32] . x2 -- fragments 32 .. 63 of the slot, 8 groups of 4, one group requested ahead.
  // GELU_OF: registers 2 g, 2 g + 1 of the PREVIOUS chunk's accumulator get bias + GELU beside group g's MFMAs.
#define PM_UP(sl, accn, GELU_STMT)                                                                                    \
  {                                                                                                                   \
    bf16x8 af[2][4];                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) af[0][q] = *reinterpret_cast<const bf16x8*>((sl) + (32 + q) * 512);  \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                                   \
      if (g < 7) {                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) af[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8*>((sl) + (32 + (g + 1) * 4 + q) * 512); \
      }                                                                                                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) accn = mfma32(af[g & 1][q], xf[g * 4 + q], accn);                  \
      GELU_STMT;                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
    }                                                                                                                 \
  }
  // register 4 q + j of the accumulator is hidden feature 32 kc + 8 q + 4 h + j
#define PM_GELU2(kc, g)                                                                                               \
  {                                                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                   \
      const int i = 2 * (g) + e;                                                                                      \
      acc1[i] = gelu_fast(acc1[i] + sbu[32 * (kc) + 8 * (i >> 2) + 4 * h + (i & 3)]);                                  \
    }                                                                                                                 \
  }
  // ---- down(kc): the 32 GELU'd hidden features as two natural-order B fragments, then 32 MFMAs into the 16 output tiles
#define PM_DOWN(sl)                                                                                                   \
  {                                                                                                                   \
    bf16x8 hf[2];                                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                \
      /* own rows: q = 2 ks -> hidden 16 ks + 4 h + (0..3), q = 2 ks + 1 -> 16 ks + 8 + 4 h + (0..3); a B fragment wants  */ \
      /* 16 ks + 8 h + (0..7): the lower lane half takes its partner's q = 2 ks rows, the upper half its partner's q = 2 ks + 1 */ \
      const bf16x4 p0 = pack4(acc1[8 * ks], acc1[8 * ks + 1], acc1[8 * ks + 2], acc1[8 * ks + 3]);                       \
      const bf16x4 p1 = pack4(acc1[8 * ks + 4], acc1[8 * ks + 5], acc1[8 * ks + 6], acc1[8 * ks + 7]);                   \
      const u32x2 v0 = __builtin_bit_cast(u32x2, p0), v1 = __builtin_bit_cast(u32x2, p1);                              \
      const auto s0 = __builtin_amdgcn_permlane32_swap(v0[0], v1[0], false, false);                                    \
      const auto s1 = __builtin_amdgcn_permlane32_swap(v0[1], v1[1], false, false);                                    \
      const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};                                                                    \
      hf[ks] = __builtin_bit_cast(bf16x8, o);                                                                          \
    }                                                                                                                 \
    bf16x8 af[2][4];                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) af[0][q] = *reinterpret_cast<const bf16x8*>((sl) + q * 512);         \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {     /* group g: k-step g >> 2, tiles 4 (g & 3) .. + 4 */            \
      if (g < 7) {                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) af[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8*>((sl) + ((g + 1) * 4 + q) * 512); \
      }                                                                                                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) acc2[4 * (g & 3) + q] = mfma32(af[g & 1][q], hf[g >> 2], acc2[4 * (g & 3) + q]); \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
    }                                                                                                                 \
  }
#define PM_ZERO(x) _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) x[i_] = 0.f

  // chunk 0: [ -- | up(0)]
  {
    PM_TOP(0);
    const bf16* sl = ring + lane * 8;
    PM_ZERO(acc1);
    PM_UP(sl, acc1, (void)0);
  }
  // chunks 1 .. 63: [down(k - 1) | up(k)] -- GELU of chunk k - 1 beside the MFMAs of up(k), then down(k - 1)
  for (int k = 1; k < 64; ++k) {
    PM_TOP(k);
    const bf16* sl = ring + (k & 1) * PM_SLOT_ELEMS + lane * 8;
    f32x16 accn;
    PM_ZERO(accn);
    PM_UP(sl, accn, PM_GELU2(k - 1, g));
    PM_DOWN(sl);
    acc1 = accn;
  }
  // chunk 64: [down(63) | -- ]; the attention rows replace x2 in the fragment registers (they land behind the next barrier)
  {
    PM_TOP(64);
    const bf16* sl = ring + lane * 8;
#pragma unroll
    for (int g = 0; g < 8; ++g) PM_GELU2(63, g);
    {
      const bf16* ap = a.AO + (long long)mc * a.ldao + 8 * h;
#pragma unroll
      for (int s = 0; s < 32; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(ap + 16 * s);
    }
    PM_DOWN(sl);
  }
  // chunks 65 .. 72: attention.dense, 4 k-steps x 16 tiles each
  for (int dc = 0; dc < 8; ++dc) {
    PM_TOP(65 + dc);
    const bf16* sl = ring + ((65 + dc) & 1) * PM_SLOT_ELEMS + lane * 8;
    bf16x8 af[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) af[0][q] = *reinterpret_cast<const bf16x8*>(sl + q * 512);
#pragma unroll
    for (int g = 0; g < 16; ++g) {                              // group g: k-step 4 dc + (g >> 2), tiles 4 (g & 3) .. + 4
      if (g < 15) {
#pragma unroll
        for (int q = 0; q < 4; ++q) af[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(sl + ((g + 1) * 4 + q) * 512);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[4 * (g & 3) + q] = mfma32(af[g & 1][q], xf[4 * dc + (g >> 2)], acc2[4 * (g & 3) + q]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: h_out = (acc + bias) + h_in (the order of k_linear's residual epilogue), then the next layer's LayerNorms with
  // k_ln_rows's arithmetic: its lane l holds features 8 l .. 8 l + 7 = group (t, u, h) here, l = 4 t + 2 u + h; its butterfly
  // folds lane bits 5 .. 0 = t bits 3 .. 0, u, h in that order
  const long long ro = (long long)m * 512;
  float gs[16][2];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f0 = 32 * t + 16 * u + 8 * h;
      const f32x4 ba = *reinterpret_cast<const f32x4*>(a.b_cat + f0), bb = *reinterpret_cast<const f32x4*>(a.b_cat + f0 + 4);
      const f32x4 ha = *reinterpret_cast<const f32x4*>(a.hin + (long long)mc * 512 + f0), hb = *reinterpret_cast<const f32x4*>(a.hin + (long long)mc * 512 + f0 + 4);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float va = (acc2[t][8 * u + j] + ba[j]) + ha[j], vb = (acc2[t][8 * u + 4 + j] + bb[j]) + hb[j];
        acc2[t][8 * u + j] = va; acc2[t][8 * u + 4 + j] = vb;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc2[t][8 * u + j];
      gs[t][u] = s;
      if (m < a.M) {
        const f32x4 oa = {acc2[t][8 * u], acc2[t][8 * u + 1], acc2[t][8 * u + 2], acc2[t][8 * u + 3]};
        const f32x4 ob = {acc2[t][8 * u + 4], acc2[t][8 * u + 5], acc2[t][8 * u + 6], acc2[t][8 * u + 7]};
        *reinterpret_cast<f32x4*>(a.hout + ro + f0) = oa;
        *reinterpret_cast<f32x4*>(a.hout + ro + f0 + 4) = ob;
      }
    }
  if (!a.nx1) return;
  // fold t bits 3, 2, 1, 0, then u, then h (the partner lane): at every level both operands are sums over the same kind of set, so
  // the value is the one every lane of k_ln_rows ends with (fp32 addition commutes)
#define PM_FOLD(gsv, tot)                                                                                             \
  {                                                                                                                   \
    _Pragma("unroll") for (int w = 8; w >= 1; w >>= 1)                                                                \
      _Pragma("unroll") for (int t = 0; t < w; ++t) { gsv[t][0] += gsv[t + w][0]; gsv[t][1] += gsv[t + w][1]; }        \
    tot = gsv[0][0] + gsv[0][1];                                                                                      \
    tot += xhalf(tot);                                                                                                \
  }
  float tot;
  PM_FOLD(gs, tot);
  const float mean = tot / 512.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d0 = acc2[t][8 * u + j] - mean; q += d0 * d0; }
      gs[t][u] = q;
    }
  float qt;
  PM_FOLD(gs, qt);
  const float rstd = rsqrtf(qt / 512.f + a.eps);
  if (m >= a.M) return;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f0 = 32 * t + 16 * u + 8 * h;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(a.g1 + f0), gb = *reinterpret_cast<const f32x4*>(a.g1 + f0 + 4);
      const f32x4 ba = *reinterpret_cast<const f32x4*>(a.b1 + f0), bb = *reinterpret_cast<const f32x4*>(a.b1 + f0 + 4);
      const f32x4 ha = *reinterpret_cast<const f32x4*>(a.g2 + f0), hb = *reinterpret_cast<const f32x4*>(a.g2 + f0 + 4);
      const f32x4 ca = *reinterpret_cast<const f32x4*>(a.b2 + f0), cb = *reinterpret_cast<const f32x4*>(a.b2 + f0 + 4);
      bf16x8 o1, o2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o1[j] = (bf16)((acc2[t][8 * u + j] - mean) * rstd * ga[j] + ba[j]); o1[4 + j] = (bf16)((acc2[t][8 * u + 4 + j] - mean) * rstd * gb[j] + bb[j]);
        o2[j] = (bf16)((acc2[t][8 * u + j] - mean) * rstd * ha[j] + ca[j]); o2[4 + j] = (bf16)((acc2[t][8 * u + 4 + j] - mean) * rstd * hb[j] + cb[j]);
      }
      *reinterpret_cast<bf16x8*>(a.nx1 + ro + f0) = o1;
      *reinterpret_cast<bf16x8*>(a.nx2 + ro + f0) = o2;
    }
}

int launch_dmlp_fused(const DMlpArgs& a, hipStream_t st) {
  if (a.M <= 0 || !a.X2 || !a.AO || a.ldao < 512 || (a.ldao % 8) || !a.hin || !a.hout || a.hin == a.hout || !a.Wm || !a.b_up || !a.b_cat ||
      (((uintptr_t)a.X2 | (uintptr_t)a.AO | (uintptr_t)a.Wm | (uintptr_t)a.hin | (uintptr_t)a.hout | (uintptr_t)a.nx1 | (uintptr_t)a.nx2) & 15) ||
      (a.nx1 && (!a.nx2 || !a.g1 || !a.b1 || !a.g2 || !a.b2)))
    ETD_FAIL(ETD_EINVAL, "dmlp_fused: bad arguments");
  ProfScope ps("k_dmlp_fused", st, 2.0 * a.M * 512.0 * (2048.0 * 2 + 512.0), (double)a.M * 512 * (2 + 2 + 4 + 4 + (a.nx1 ? 4 : 0)) + (double)DMLP_STREAM_ELEMS * 2);
  hipLaunchKernelGGL(k_dmlp_fused, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// Host side: dense_h_to_4h [2048][512] and the K-concatenated (dense_4h_to_h | attention.dense) [512][2560], both already bf16 in
// nn.Linear layout -> the kernel's stream: DMLP_NCHUNK chunks of 64 fragments x 64 lanes x 8 elements.
//   chunk k = 0 .. 64:  fragments  0 .. 31 = down(k - 1): (k-step ks, tile t) at 16 ks + t (zeros for k = 0)
//                       fragments 32 .. 63 = up(k): k-step s at 32 + s                     (zeros for k = 64)
//   chunk 65 + dc:      attention.dense, (k-step 4 dc + sl, tile t) at 16 sl + t
void pack_dmlp_weights(const uint16_t* Wup, const uint16_t* Wcat, uint16_t* dst) {
  auto perm = [](int r) { const int i = (r & 3) + 4 * (r >> 3), hh = (r >> 2) & 1; return 16 * (i >> 3) + 8 * hh + (i & 7); };   // A-row r -> feature offset in its 32-tile
  const int KC = 2560;
  for (int k = 0; k < DMLP_NCHUNK; ++k)
    for (int fr = 0; fr < 64; ++fr)
      for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        uint16_t* d = dst + (((size_t)k * 64 + fr) * 64 + l) * 8;
        for (int j = 0; j < 8; ++j) d[j] = 0;
        if (k <= 64) {
          if (fr < 32) {
            if (k == 0) continue;
            const int ks = fr >> 4, t = fr & 15, feat = 32 * t + perm(r);
            for (int j = 0; j < 8; ++j) d[j] = Wcat[(size_t)feat * KC + 32 * (k - 1) + 16 * ks + 8 * h + j];
          } else {
            if (k == 64) continue;
            const int s = fr - 32;
            for (int j = 0; j < 8; ++j) d[j] = Wup[(size_t)(32 * k + r) * 512 + 16 * s + 8 * h + j];
          }
        } else {
          const int dc = k - 65, sl = fr >> 4, t = fr & 15, feat = 32 * t + perm(r), s = 4 * dc + sl;
          for (int j = 0; j < 8; ++j) d[j] = Wcat[(size_t)feat * KC + 2048 + 16 * s + 8 * h + j];
        }
      }
}
