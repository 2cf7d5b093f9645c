// Batched prefill of the EtudeDecoder, d16: the MLP branch, attention.dense, the parallel residual and the NEXT layer's two
// LayerNorms of a GPT-NeoX layer in ONE launch                      modeling_gpt_neox.py:239-245 (mlp), :250-272 (layer)
//
//   h_out = h_in + [W2 gelu(W1 x2 + b1) + Wd attn + (b2 + bd)],   x1' = LN1'(h_out),  x2' = LN2'(h_out)
//
// The default path is three launches per layer (up + GELU -> Xcat, (down | dense) + residual, LayerNorm rows) that move 18 KB per
// token through HBM and run their 128 x 256 tiles at 460-520 TFLOP/s -- bound by L2 -> CU traffic (52 FLOP per byte the tile pulls
// from L2).  Here the token tile stays in registers for the whole branch, as in the extractor's k_ffn_fused.
//
// STATUS: on by default since the end of round 2 (ETD_FUSED_PMLP=0 turns it off); bit-identical to the three-launch path (logits of a 600-token prompt, token
// streams of 48 batched jobs: tests/test_gpu_decoder.py; first tokens and K/V cache row sums of a 256 k-row pass: tools/bench_prefill.py --digest).
// Per 256 500-row launch: round 3 1.70 ms (711 TFLOP/s), round 4 **1.49 ms** (810).  What each step was, in order:
//   round 2 (every step kept the bit-identity test green, and every one is needed):
//   1. a token's input fragments (128 registers) and 512 fp32 outputs (256) leave 128 of a wave's 512 registers: hipcc put all
//      256 accumulator registers into AGPRs, the fragments into VGPRs and copied them through ONE AGPR quad (272 v_accvgpr moves
//      per 64-MFMA chunk); with the fragments loaded straight into AGPRs (global_load ... a[n:n+3]) and read from there as the
//      MFMA's B operand, and the output tiles split 8 AGPR / 8 VGPR by asm constraints, no copy is left;
//   2. hipcc read every weight fragment right in front of its MFMA (s_waitcnt lgkmcnt(0) per MFMA: with ONE wave per SIMD nobody
//      hides that): the fragment reads, counted waits and MFMAs are volatile asm in program order, one 4-fragment group ahead
//      (two groups ahead spills 19 registers, and a scratch reload's vmcnt(0) drains the LDS-DMA);
//   3. the bias values of the GELU are read by hand too (a compiler-placed ds_read brings lgkmcnt(0) and drains the prefetch);
//   4. the GELU halves are pinned BETWEEN the chained MFMAs (empty volatile asms): LLVM otherwise sinks all sixteen behind the up
//      phase, where they block the in-order issue for ~1100 clocks per chunk;
//   round 4 (phase stamps: tools/bench_prefill.py --stamps on a -DETD_PMLP_STAMP build; a workgroup lived 428 k clocks of which 131 k were its epilogue):
//   5. chunks 1 .. 63 as 16 uniform groups -- one GELU and at most two LDS-DMA pieces per 4 MFMAs, the sixteen GELUs of a chunk spread over four phases, the DMA
//      pieces of the next chunk issued in groups 0 .. 11 with one address per four (pm_group / pm_chunk below): 4 623 -> 3 707 clocks per chunk;
//   6. the epilogue's rows through the idle ring as full 512-byte segments (token-on-lane accesses touched 32 rows x 32 bytes per instruction) and its five
//      parameter vectors from LDS: 131 k -> 88 k clocks.
// What is left of a workgroup's 389 k clocks: chunks 1 .. 63 235 k (3 730 per chunk against 2 048 of MFMAs: one wave per SIMD issues everything in order, and GELU
// alone is 1 216 clocks of VALU per chunk), the token fragments' loads 18 k + 10 k (x2, attention rows: fragment-shaped), chunk 64 + the dense chunks 42 k, epilogue 88 k
// (768 KB per workgroup: 12 B / clk per CU with every CU streaming).  A dead end on the way: the DMA issue BETWEEN the independent MFMAs of a down group gave wrong
// results in round 2 (a register the compiler took for free was still the A operand of a queued MFMA) -- the asm form is only safe where the compiler's reuse
// distance is known; starting a launch's first round of workgroups staggered (s_sleep by blockIdx & 7) only added the idle time (round 4).
//   * a wave owns 32 tokens; x2 enters once as the 32 B-operand fragments of v_mfma_f32_32x32x16_bf16 (128 registers);
//   * the 2048-wide hidden layer exists 32 features at a time: acc1 = W1[32 rows] . x2 (32 chained MFMAs), bias + erf-GELU +
//     d16 rounding in registers; two v_permlane32_swap per k-step turn the accumulator's row order into the natural k order of a
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

#include <cstdlib>
#include <utility>
#define PM_SLOT_ELEMS (32 * 1024)          // d16 elements per ring slot: 64 fragments of 512 elements (64 KiB)

typedef const __attribute__((address_space(1))) void* pm_gptr_t;
typedef __attribute__((address_space(3))) void* pm_lptr_t;

// ---- hand-placed instruction stream.  One wave per SIMD has nobody to hide an LDS read behind, and hipcc, short of registers,
// reads each weight fragment right in front of the MFMA that uses it (s_waitcnt lgkmcnt(0) per MFMA: ~24 % MFMA duty).  So the
// fragment reads, their counted waits and the MFMAs are volatile asm statements in program order: a chunk's 64 fragments are 16
// groups of 4, group j + 1 is requested before group j is waited for (lgkmcnt(4): the younger group stays in flight; LDS
// returns in order, and a read the compiler adds of its own only makes the wait stricter), two 4-fragment buffers alternate
// (a third, two groups ahead, does not fit: 19 registers spilled and every reload drained the LDS-DMA with vmcnt(0)).
// Register classes are explicit too: the 32 x2 / attention fragments are BORN in AGPRs (a global load may name an AccVGPR
// destination) and are read from there as the MFMA's B operand; output tiles 0..7 live in AGPRs, 8..15 in VGPRs -- 256 + 241.
// (Left to itself hipcc puts all 256 accumulator registers into AGPRs and shuttles the fragments through one AGPR quad.)
template <int F> __device__ __forceinline__ void pm_rd(d16x8& b, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(F * 1024));
}
template <int F0> __device__ __forceinline__ void pm_rd4(d16x8 (&b)[4], unsigned addr) {
  pm_rd<F0>(b[0], addr); pm_rd<F0 + 1>(b[1], addr); pm_rd<F0 + 2>(b[2], addr); pm_rd<F0 + 3>(b[3], addr);
}
template <int N> __device__ __forceinline__ void pm_wait_lds() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// acc (VGPR) += A (VGPR) . B (AGPR)
__device__ __forceinline__ void pm_mfma_v_a(f32x16& acc, const d16x8& af, const d16x8& bf) {
  asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+v"(acc) : "v"(af), "a"(bf));
}
// output tile T (AGPR for T < 8, VGPR above) += A (VGPR) . B (VGPR: the hidden fragment / AGPR: an attention fragment)
template <int T, bool B_AGPR> __device__ __forceinline__ void pm_mfma_out(f32x16& acc, const d16x8& af, const d16x8& bf) {
  if constexpr (T < 8) {
    if constexpr (B_AGPR) asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+a"(acc) : "v"(af), "a"(bf));
    else                  asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+a"(acc) : "v"(af), "v"(bf));
  } else {
    if constexpr (B_AGPR) asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+v"(acc) : "v"(af), "a"(bf));
    else                  asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+v"(acc) : "v"(af), "v"(bf));
  }
}
// The NEXT chunk's 16 one-KiB LDS-DMA pieces of this wave are issued one per fragment group (two in the half chunks), behind the
// group's MFMAs: issued as one burst at the chunk top they cost the wave ~100 clocks each with an idle MFMA pipe.
struct PmNext { const d16* src; d16* dst; bool on; };
template <int I> __device__ __forceinline__ void pm_dma(const PmNext& n) {
  // four pieces share one address computation: the instruction's immediate offset serves the global and the LDS address alike
  if (n.on) __builtin_amdgcn_global_load_lds((pm_gptr_t)(n.src + (I >> 2) * 2048), (pm_lptr_t)(n.dst + (I >> 2) * 2048), 16, (I & 3) * 1024, 0);
}
// fragments of group J of an MLP chunk [down(k - 1): 0 .. 31 | up(k): 32 .. 63]: groups 0 .. 7 = up, 8 .. 15 = down
template <int J> struct PmMlpGroup { static constexpr int f0 = J < 8 ? 32 + 4 * J : 4 * (J - 8); };
template <int J, int JEND> __device__ __forceinline__ void pm_roll(d16x8 (&buf)[2][4], unsigned addr) {
  // in front of group J: request group J + 1 (if the chunk has one), then wait until group J has landed
  if constexpr (J + 1 < JEND) { pm_rd4<PmMlpGroup<J + 1>::f0>(buf[(J + 1) & 1], addr); pm_wait_lds<4>(); }
  else pm_wait_lds<0>();
}
// register 4 q + j of the hidden accumulator is hidden feature 32 kc + 8 q + 4 h + j: bias + erf-GELU of registers 2 J, 2 J + 1
template <int J> __device__ __forceinline__ void pm_gelu2(f32x16& acc1, const float* sbu, int kc, int h) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    constexpr int dummy = 0; (void)dummy;
    const int i = 2 * J + e;
    acc1[i] = gelu_fast(acc1[i] + sbu[32 * kc + 8 * (i >> 2) + 4 * h + (i & 3)]);
  }
}
// up group J (0 .. 7) of chunk 0 into accn (no GELU beside it: there is no previous chunk)
template <int J> __device__ __forceinline__ void pm_up_group0(d16x8 (&buf)[2][4], unsigned addr, f32x16& accn, const d16x8 (&xf)[32], const PmNext& nx) {
  pm_roll<J, 8>(buf, addr);
#pragma unroll
  for (int q = 0; q < 4; ++q) pm_mfma_v_a(accn, buf[J & 1][q], xf[4 * J + q]);
  pm_dma<2 * J>(nx); pm_dma<2 * J + 1>(nx);
  __builtin_amdgcn_sched_barrier(0);
}
// down group J (8 .. 15): k-step (J - 8) >> 2, tiles 4 ((J - 8) & 3) .. + 4
template <int J, bool HALF> __device__ __forceinline__ void pm_down_group(d16x8 (&buf)[2][4], unsigned addr, f32x16 (&acc2)[16], const d16x8 (&hf)[2], const PmNext& nx) {
  pm_roll<J, 16>(buf, addr);
  constexpr int g = J - 8, t0 = 4 * (g & 3);
  pm_mfma_out<t0 + 0, false>(acc2[t0 + 0], buf[J & 1][0], hf[g >> 2]);
  pm_mfma_out<t0 + 1, false>(acc2[t0 + 1], buf[J & 1][1], hf[g >> 2]);
  pm_mfma_out<t0 + 2, false>(acc2[t0 + 2], buf[J & 1][2], hf[g >> 2]);
  pm_mfma_out<t0 + 3, false>(acc2[t0 + 3], buf[J & 1][3], hf[g >> 2]);
  if constexpr (HALF) { pm_dma<2 * g>(nx); pm_dma<2 * g + 1>(nx); } else pm_dma<J>(nx);
  __builtin_amdgcn_sched_barrier(0);
}
// attention.dense group J (0 .. 15) of dense chunk DC: fragments 4 J .. + 4 = k-step 4 DC + (J >> 2), tiles 4 (J & 3) .. + 4
template <int DC, int J> __device__ __forceinline__ void pm_dense_group(d16x8 (&buf)[2][4], unsigned addr, f32x16 (&acc2)[16], const d16x8 (&xf)[32], const PmNext& nx) {
  if constexpr (J + 1 < 16) { pm_rd4<4 * (J + 1)>(buf[(J + 1) & 1], addr); pm_wait_lds<4>(); }
  else pm_wait_lds<0>();
  constexpr int t0 = 4 * (J & 3), s = 4 * DC + (J >> 2);
  pm_mfma_out<t0 + 0, true>(acc2[t0 + 0], buf[J & 1][0], xf[s]);
  pm_mfma_out<t0 + 1, true>(acc2[t0 + 1], buf[J & 1][1], xf[s]);
  pm_mfma_out<t0 + 2, true>(acc2[t0 + 2], buf[J & 1][2], xf[s]);
  pm_mfma_out<t0 + 3, true>(acc2[t0 + 3], buf[J & 1][3], xf[s]);
  pm_dma<J>(nx);
  __builtin_amdgcn_sched_barrier(0);
}
template <int DC, int... Js> __device__ __forceinline__ void pm_dense_chunk(d16x8 (&buf)[2][4], unsigned addr, f32x16 (&acc2)[16], const d16x8 (&xf)[32], const PmNext& nx,
                                                                            std::integer_sequence<int, Js...>) {
  pm_rd4<0>(buf[0], addr);
  (pm_dense_group<DC, Js>(buf, addr, acc2, xf, nx), ...);
}
// the 32 GELU'd hidden features of a chunk as two natural-order B fragments: own rows are q = 2 ks -> hidden 16 ks + 4 h + (0..3)
// and q = 2 ks + 1 -> 16 ks + 8 + 4 h + (0..3); a B fragment wants 16 ks + 8 h + (0..7): the lower lane half takes its partner's
// q = 2 ks rows, the upper half its partner's q = 2 ks + 1
__device__ __forceinline__ void pm_hidden_frags(const f32x16& acc1, d16x8 (&hf)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const d16x4 p0 = pack4d(acc1[8 * ks], acc1[8 * ks + 1], acc1[8 * ks + 2], acc1[8 * ks + 3]);
    const d16x4 p1 = pack4d(acc1[8 * ks + 4], acc1[8 * ks + 5], acc1[8 * ks + 6], acc1[8 * ks + 7]);
    const u32x2 v0 = __builtin_bit_cast(u32x2, p0), v1 = __builtin_bit_cast(u32x2, p1);
    const auto s0 = __builtin_amdgcn_permlane32_swap(v0[0], v1[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(v0[1], v1[1], false, false);
    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
    hf[ks] = __builtin_bit_cast(d16x8, o);
  }
  // (VALU -> MFMA source wait states: the compiler does not know the asm below is an MFMA.  The fragments are operands of the nop so
  // that the arithmetic producing them cannot be scheduled behind it.)
  asm volatile("s_nop 4" : "+v"(hf[0]), "+v"(hf[1]) : : "memory");
}
template <int... Js> __device__ __forceinline__ void pm_up_only(d16x8 (&buf)[2][4], unsigned addr, f32x16& accn, const d16x8 (&xf)[32], const PmNext& nx, std::integer_sequence<int, Js...>) {
  pm_rd4<PmMlpGroup<0>::f0>(buf[0], addr);
  (pm_up_group0<Js>(buf, addr, accn, xf, nx), ...);
}
template <int... Js> __device__ __forceinline__ void pm_down_only(d16x8 (&buf)[2][4], unsigned addr, f32x16 (&acc2)[16], const d16x8 (&hf)[2], const PmNext& nx, std::integer_sequence<int, Js...>) {
  // chunk 64: only the down half exists -- groups 8 .. 15 with their own lead
  pm_rd4<PmMlpGroup<8>::f0>(buf[8 & 1], addr);
  (pm_down_group<8 + Js, true>(buf, addr, acc2, hf, nx), ...);
}

// ---- round 4: chunks 1 .. 63 as 16 UNIFORM groups.  Phase stamps of the round-3 schedule (tools/bench_prefill.py --stamps, clocks per chunk of 2 048 MFMA clocks):
// up phase + GELU 2 264, down phase 1 382, wait for the chunk's LDS-DMA 715, barrier 138, hidden fragments 124 = 4 623.  One wave per SIMD issues in order, an MFMA
// leaves ~24 of its 32 clocks for other issue, and the round-3 stream put all sixteen GELUs (~76 issue clocks each) into the 32 gaps of the up phase while the down
// phase's gaps held one DMA and four LDS reads; the last DMA piece of chunk k + 1 was issued ~130 clocks before the wait for it.  Now:
//   * every group is 4 MFMAs + ONE GELU (halves behind MFMA 0 and MFMA 1) + at most two DMA pieces (behind MFMAs 2 and 3).  The sixteen GELUs of up(k - 1)'s
//     accumulator are spread over FOUR phases: registers 0 .. 3 in the last four groups of chunk k - 1 itself (down k-step 1: the up accumulator has been complete
//     since group 7), 4 .. 7 beside up(k) k-steps 0 .. 15, 8 .. 11 beside k-steps 16 .. 31, 12 .. 15 beside down(k - 1) k-step 0; the hidden fragment of k-step 0
//     (registers 0 .. 7) is packed behind group 7, that of k-step 1 behind group 11.  Same operations on the same values in the same order per element, and per
//     accumulator tile the same MFMA order: bit-identical to the round-3 schedule (tests/test_gpu_decoder.py, tools/bench_prefill.py --digest);
//   * the 16 DMA pieces of chunk k + 1 leave in groups 0 .. 11 (2, 2, 1, 1 | 2, 2, 1, 1 | 1, 1, 1, 1): the last one has the whole k-step-1 phase to land;
//   * four pieces share one address computation (the instruction's immediate offset serves the global and the LDS address alike);
//   * the two up accumulators swap roles from chunk to chunk (no 16-register copy), and the first MFMA of a chain takes C = 0 (no zero fill).
// acc (VGPR) = A (VGPR) . B (AGPR): the first MFMA of an up chain
__device__ __forceinline__ void pm_mfma_v_a0(f32x16& acc, const d16x8& af, const d16x8& bf) {
  asm volatile(ETD_MFMA16_DEC " %0, %1, %2, 0" : "=&v"(acc) : "v"(af), "a"(bf));      // early clobber: the 16 result registers must not overlap the A operand
}
// group J of chunk [down(k - 1): groups 8 .. 15 | up(k): groups 0 .. 7]; GR >= 0: GELU of register GR of `g` (bias word at bias_addr + its offset); ND pieces from D0
template <int J, int GR, int D0, int ND>
__device__ __forceinline__ void pm_group(d16x8 (&buf)[2][4], unsigned addr, f32x16& accu, f32x16 (&acc2)[16], const d16x8 (&xf)[32], const d16x8& hfk,
                                         f32x16& g, unsigned bias_addr, const PmNext& nx) {
  constexpr bool UP = J < 8, GELU = GR >= 0, LAST = J == 15;
  constexpr int g8 = UP ? 0 : J - 8, t0 = 4 * (g8 & 3);
  float bb = 0.f;
  if constexpr (GELU) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bb) : "v"(bias_addr), "n"((8 * ((GR < 0 ? 0 : GR) >> 2) + ((GR < 0 ? 0 : GR) & 3)) * 4));
  // request group J + 1 (the bias word went first: LDS returns in order), then wait until group J has landed
  if constexpr (!LAST) { pm_rd4<PmMlpGroup<J + 1>::f0>(buf[(J + 1) & 1], addr); pm_wait_lds<GELU ? 5 : 4>(); }
  else pm_wait_lds<GELU ? 1 : 0>();
#define PM_PIN(x) asm volatile("" : "+v"(x))
#define PM_MFMA(q)                                                                                                              \
  do {                                                                                                                          \
    if constexpr (UP) { if constexpr (J == 0 && (q) == 0) pm_mfma_v_a0(accu, buf[J & 1][q], xf[4 * J + (q)]); else pm_mfma_v_a(accu, buf[J & 1][q], xf[(UP ? 4 * J : 0) + (q)]); } \
    else pm_mfma_out<t0 + (q), false>(acc2[t0 + (q)], buf[J & 1][q], hfk);                                                      \
  } while (0)
  PM_MFMA(0);
  // gelu_fast (dec_epilogue.h) by hand, the same operations in the same order, one half behind each of the group's first two MFMAs
#ifdef ETD_GELU_AS
  float x0 = 0.f, tt = 0.f, ee = 0.f;
  if constexpr (GELU) {
    // the bias word has landed (nothing younger than group J + 1's fragments is outstanding); bb is an operand of the wait so that no use of it moves above
    if constexpr (!LAST) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bb) : : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb) : : "memory");
    x0 = g[GR < 0 ? 0 : GR] + bb;
    const float z0 = fabsf(x0) * 0.70710678118654752440f;
    tt = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z0, 1.f));
    ee = __builtin_amdgcn_exp2f(-1.4426950408889634f * z0 * z0);
    PM_PIN(x0); PM_PIN(tt); PM_PIN(ee);
  }
  PM_MFMA(1);
  if constexpr (GELU) {
    PM_PIN(x0); PM_PIN(tt); PM_PIN(ee);
    float p = fmaf(1.061405429f, tt, -1.453152027f);
    p = fmaf(p, tt, 1.421413741f); p = fmaf(p, tt, -0.284496736f); p = fmaf(p, tt, 0.254829592f);
    const float er = fmaf(-p * tt, ee, 1.f);
    float r0 = 0.5f * x0 * (1.f + copysignf(er, x0));
    PM_PIN(r0);
    g[GR < 0 ? 0 : GR] = r0;
  }
#else
  float x0 = 0.f, uu = 0.f, ss = 0.f, qq = 0.f;
  if constexpr (GELU) {
    // the bias word has landed (nothing younger than group J + 1's fragments is outstanding); bb is an operand of the wait so that no use of it moves above
    if constexpr (!LAST) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bb) : : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb) : : "memory");
    x0 = g[GR < 0 ? 0 : GR] + bb;
    uu = __builtin_amdgcn_fmed3f(x0, -3.875f, 3.875f);
    ss = uu * uu;
    qq = fmaf(3.1433767589e-08f, ss, -2.0315644633e-06f);
    qq = fmaf(qq, ss, 5.6378066802e-05f);
    PM_PIN(x0); PM_PIN(uu); PM_PIN(ss); PM_PIN(qq);
  }
  PM_MFMA(1);
  if constexpr (GELU) {
    PM_PIN(x0); PM_PIN(uu); PM_PIN(ss); PM_PIN(qq);
    qq = fmaf(qq, ss, -8.9401804144e-04f); qq = fmaf(qq, ss, 9.1527355835e-03f);
    qq = fmaf(qq, ss, -6.5392248333e-02f); qq = fmaf(qq, ss, 3.9845609665e-01f);
    float r0 = x0 * fmaf(qq, uu, 0.5f);
    PM_PIN(r0);
    g[GR < 0 ? 0 : GR] = r0;
  }
#endif
  PM_MFMA(2);
  if constexpr (ND >= 1) pm_dma<D0>(nx);
  PM_MFMA(3);
  if constexpr (ND >= 2) pm_dma<D0 + 1>(nx);
#undef PM_MFMA
#undef PM_PIN
  __builtin_amdgcn_sched_barrier(0);
}
// the GELU'd hidden features of k-step KS (registers 8 KS .. 8 KS + 7) as the natural-order B fragment (pm_hidden_frags, one k-step)
template <int KS> __device__ __forceinline__ void pm_hidden_frag1(const f32x16& acc1, d16x8& hfk) {
  const d16x4 p0 = pack4d(acc1[8 * KS], acc1[8 * KS + 1], acc1[8 * KS + 2], acc1[8 * KS + 3]);
  const d16x4 p1 = pack4d(acc1[8 * KS + 4], acc1[8 * KS + 5], acc1[8 * KS + 6], acc1[8 * KS + 7]);
  const u32x2 v0 = __builtin_bit_cast(u32x2, p0), v1 = __builtin_bit_cast(u32x2, p1);
  const auto s0 = __builtin_amdgcn_permlane32_swap(v0[0], v1[0], false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap(v0[1], v1[1], false, false);
  const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
  hfk = __builtin_bit_cast(d16x8, o);
  asm volatile("s_nop 4" : "+v"(hfk) : : "memory");      // (VALU -> MFMA source wait states: the compiler does not know the asm below is an MFMA)
}
// one chunk k in 1 .. 63.  prev = up(k - 1)'s accumulator (registers 0 .. 3 already GELU'd), cur = up(k)'s (written here; its registers 0 .. 3 leave GELU'd)
// AOL (chunk 63 only): the attention rows that replace x2 in the fragment registers for attention.dense are requested HERE, four fragments behind each of the eight down groups --
// x2's last use is group 7 -- instead of at the top of chunk 64, where the 32 fragment-shaped loads + their wait cost a wave ~8 k clocks with an idle MFMA pipe
#define PM_LOADX(dst, ptr, S) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(ptr), "n"(32 * (S)) : "memory")      /* fragment S of the lane's row: one base register, immediate offsets */
template <bool AOL>
__device__ __forceinline__ void pm_chunk(d16x8 (&buf)[2][4], unsigned sa, f32x16& prev, f32x16& cur, f32x16 (&acc2)[16], d16x8 (&xf)[32], d16x8 (&hf)[2],
                                         unsigned bias_prev, unsigned bias_cur, const PmNext& nx, const d16* ao, long long ao_off) {
  const d16* ap = nullptr;
  if constexpr (AOL) { ap = ao + ao_off; asm volatile("" : "+v"(ap)); }      // (formed here, not held through the 62 chunks before)
#define PM_AOL(G) do { if constexpr (AOL) { PM_LOADX(xf[4 * (G)], ap, 4 * (G)); PM_LOADX(xf[4 * (G) + 1], ap, 4 * (G) + 1); PM_LOADX(xf[4 * (G) + 2], ap, 4 * (G) + 2); PM_LOADX(xf[4 * (G) + 3], ap, 4 * (G) + 3); } } while (0)
  pm_rd4<PmMlpGroup<0>::f0>(buf[0], sa);
  pm_group<0, 4, 0, 2>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<1, 5, 2, 2>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<2, 6, 4, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<3, 7, 5, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<4, 8, 6, 2>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<5, 9, 8, 2>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<6, 10, 10, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_group<7, 11, 11, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  pm_hidden_frag1<0>(prev, hf[0]);
  pm_group<8, 12, 12, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  PM_AOL(0);
  pm_group<9, 13, 13, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  PM_AOL(1);
  pm_group<10, 14, 14, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  PM_AOL(2);
  pm_group<11, 15, 15, 1>(buf, sa, cur, acc2, xf, hf[0], prev, bias_prev, nx);
  PM_AOL(3);
  pm_hidden_frag1<1>(prev, hf[1]);
  pm_group<12, 0, 0, 0>(buf, sa, cur, acc2, xf, hf[1], cur, bias_cur, nx);
  PM_AOL(4);
  pm_group<13, 1, 0, 0>(buf, sa, cur, acc2, xf, hf[1], cur, bias_cur, nx);
  PM_AOL(5);
  pm_group<14, 2, 0, 0>(buf, sa, cur, acc2, xf, hf[1], cur, bias_cur, nx);
  PM_AOL(6);
  pm_group<15, 3, 0, 0>(buf, sa, cur, acc2, xf, hf[1], cur, bias_cur, nx);
  PM_AOL(7);
#undef PM_AOL
}

// Diagnostic build (-DETD_PMLP_STAMP, tools/bench_prefill.py --stamps): s_memtime at the phase boundaries of chunks 1 .. 63, summed per wave in SGPRs
// (workgroups 0 .. 63) and read back with etd_debug_pmlp_stamps: [0] the chunk's 16 groups, [1] wait for the chunk's LDS-DMA, [2] barrier (sums over chunks 1 .. 63); [3] prologue + chunk 0, [4] chunks 1 .. 63, [5] chunks 64 .. 72,
// [6] residual + h_out, [7] the wave's whole life
// (round 3's schedule: [0] down phase, [3] up phase + GELU, [4] hidden fragments).  The shipped build has no stamp.
#ifdef ETD_PMLP_STAMP
__device__ unsigned long long g_pmlp_stamp[64 * 4 * 8];
#define PMS(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ps_acc[i] += t_ - ps_t; ps_t = t_; } while (0)
#else
#define PMS(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dmlp_fused(DMlpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PM_SLOT_ELEMS * 2 + 2048 * 4];
  d16* ring = reinterpret_cast<d16*>(smem);
  float* sbu = reinterpret_cast<float*>(smem + 2 * PM_SLOT_ELEMS * 2);      // b_up[2048]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;
  const unsigned ring_lds = (unsigned)reinterpret_cast<uintptr_t>(ring) + lane * 16;      // LDS byte address of this lane's 16 bytes of fragment 0, slot 0
#ifdef ETD_PMLP_STAMP
  const unsigned long long ps_t0 = __builtin_amdgcn_s_memtime();
#endif
  const unsigned sbu_lds = (unsigned)reinterpret_cast<uintptr_t>(sbu) + h * 16;            // ... of b_up[4 h]: register 4 q + j of a hidden accumulator wants b_up[32 kc + 8 q + 4 h + j]

  // ring slot (k & 1) <- stream chunk k: 64 one-KiB pieces, 16 per wave
  auto issue = [&](int k) {
    const d16* src = a.Wm + (long long)k * PM_SLOT_ELEMS + wave * (16 * 512) + lane * 8;
    d16* dst = ring + (k & 1) * PM_SLOT_ELEMS + wave * (16 * 512);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      __builtin_amdgcn_global_load_lds((pm_gptr_t)(src + i * 512), (pm_lptr_t)(dst + i * 512), 16, 0, 0);
  };
#define PM_TOP(k)                                                                                                     \
  do {                                                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* this wave's pieces of chunk k have landed ... */   \
    __syncthreads();                                    /* ... and everybody's; every wave is done reading the other slot */ \
  } while (0)
  // this wave's 16 pieces of chunk k (source, destination slot); nothing behind the last chunk
  auto next_of = [&](int k) {
    return PmNext{a.Wm + (long long)(k + 1) * PM_SLOT_ELEMS + wave * (16 * 512) + lane * 8, ring + ((k + 1) & 1) * PM_SLOT_ELEMS + wave * (16 * 512), k + 1 < DMLP_NCHUNK};
  };

  issue(0);
  // the token tile as B fragments: lane (token r, half h) holds x2[token][16 s + 8 h .. + 8], s = 0 .. 31 -- loaded straight into
  // AGPRs; the compiler does not count these loads, PM_TOP(0)'s vmcnt(0) does
  d16x8 xf[32];
  {
    const d16* xp = a.X2 + (long long)mc * 512 + 8 * h;
#define PM_X4(S) PM_LOADX(xf[S], xp, S); PM_LOADX(xf[(S) + 1], xp, (S) + 1); PM_LOADX(xf[(S) + 2], xp, (S) + 2); PM_LOADX(xf[(S) + 3], xp, (S) + 3)
    PM_X4(0); PM_X4(4); PM_X4(8); PM_X4(12); PM_X4(16); PM_X4(20); PM_X4(24); PM_X4(28);
#undef PM_X4
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#ifdef ETD_PMLP_STAMP
  const unsigned long long ps_tx = __builtin_amdgcn_s_memtime();      // token fragments loaded
#endif
  for (int i = tid; i < 2048; i += 256) sbu[i] = a.b_up[i];

  f32x16 acc2[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
  f32x16 accA, accB;
  d16x8 buf[2][4], hf[2];
  using seq8 = std::make_integer_sequence<int, 8>;
  using seq16 = std::make_integer_sequence<int, 16>;
#define PM_ZERO(x) _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) x[i_] = 0.f

  // chunk 0: [ -- | up(0)], then the GELU of its registers 0 .. 3 (what the last four groups of every later chunk do for their own up accumulator)
  {
    PM_TOP(0);
    PM_ZERO(accA);
    pm_up_only(buf, ring_lds, accA, xf, next_of(0), seq8{});
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accA) : : "memory");      // (MFMA -> VALU wait states: the GELU reads the accumulator)
    pm_gelu2<0>(accA, sbu, 0, h); pm_gelu2<1>(accA, sbu, 0, h);
  }
#ifdef ETD_PMLP_STAMP
  unsigned long long ps_acc[5] = {0, 0, 0, 0, 0}, ps_t = __builtin_amdgcn_s_memtime();
  const unsigned long long ps_t1 = ps_t;          // prologue + chunk 0 done
#define PM_TOP_S(k)                                                                         \
  do {                                                                                      \
    if ((k) > 1) PMS(0); else ps_t = __builtin_amdgcn_s_memtime();                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                             \
    PMS(1);                                                                                 \
    __syncthreads();                                                                        \
    PMS(2);                                                                                 \
  } while (0)
#else
#define PM_TOP_S(k) PM_TOP(k)
#endif
  // chunks 1 .. 63: [down(k - 1) | up(k)], 16 uniform groups each (pm_chunk); the two up accumulators alternate
#define PM_CHUNK(k, prev, cur, AOLV)                                                                                                                \
  do {                                                                                                                                          \
    PM_TOP_S(k);                                                                                                                                \
    pm_chunk<(AOLV)>(buf, ring_lds + ((k) & 1) * (PM_SLOT_ELEMS * 2), prev, cur, acc2, xf, hf, sbu_lds + ((k) - 1) * 128, sbu_lds + (k) * 128, next_of(k), a.AO, (long long)mc * a.ldao + 8 * h); \
  } while (0)
  for (int k = 1; k < 63; k += 2) {
    PM_CHUNK(k, accA, accB, false);
    PM_CHUNK(k + 1, accB, accA, false);
  }
  PM_CHUNK(63, accA, accB, true);
#ifdef ETD_PMLP_STAMP
  PMS(0);
  const unsigned long long ps_t2 = ps_t;          // chunk 63 done
#endif
  // chunk 64: [down(63) | -- ]; registers 4 .. 15 of up(63) still want their GELU; the attention rows replace x2 in the fragment registers
  {
    PM_TOP(64);
    const unsigned sa = ring_lds;
    pm_gelu2<2>(accB, sbu, 63, h); pm_gelu2<3>(accB, sbu, 63, h);
    pm_gelu2<4>(accB, sbu, 63, h); pm_gelu2<5>(accB, sbu, 63, h); pm_gelu2<6>(accB, sbu, 63, h); pm_gelu2<7>(accB, sbu, 63, h);
    // (the attention rows were requested behind the down groups of chunk 63; PM_TOP(64)'s vmcnt(0) has waited for them)
    pm_hidden_frags(accB, hf);
    pm_down_only(buf, sa, acc2, hf, next_of(64), seq8{});
  }
#ifdef ETD_PMLP_STAMP
  const unsigned long long ps_t2b = __builtin_amdgcn_s_memtime();     // chunk 64 done
#endif
  // chunks 65 .. 72: attention.dense, 4 k-steps x 16 tiles each
  { PM_TOP(65); pm_dense_chunk<0>(buf, ring_lds + (65 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(65), seq16{}); }
  { PM_TOP(66); pm_dense_chunk<1>(buf, ring_lds + (66 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(66), seq16{}); }
  { PM_TOP(67); pm_dense_chunk<2>(buf, ring_lds + (67 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(67), seq16{}); }
  { PM_TOP(68); pm_dense_chunk<3>(buf, ring_lds + (68 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(68), seq16{}); }
  { PM_TOP(69); pm_dense_chunk<4>(buf, ring_lds + (69 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(69), seq16{}); }
  { PM_TOP(70); pm_dense_chunk<5>(buf, ring_lds + (70 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(70), seq16{}); }
  { PM_TOP(71); pm_dense_chunk<6>(buf, ring_lds + (71 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(71), seq16{}); }
  { PM_TOP(72); pm_dense_chunk<7>(buf, ring_lds + (72 & 1) * (PM_SLOT_ELEMS * 2), acc2, xf, next_of(72), seq16{}); }

#ifdef ETD_PMLP_STAMP
  const unsigned long long ps_t3 = __builtin_amdgcn_s_memtime();      // chunks 64 .. 72 done
#endif
  // (MFMA -> VALU wait states behind the asm MFMAs; the tiles are operands, so no read of them can be scheduled before the nops)
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc2[0]), "+a"(acc2[1]), "+a"(acc2[2]), "+a"(acc2[3]), "+a"(acc2[4]), "+a"(acc2[5]), "+a"(acc2[6]), "+a"(acc2[7]) : : "memory");
  asm volatile("" : "+v"(acc2[8]), "+v"(acc2[9]), "+v"(acc2[10]), "+v"(acc2[11]), "+v"(acc2[12]), "+v"(acc2[13]), "+v"(acc2[14]), "+v"(acc2[15]));
  // ---- epilogue: h_out = (acc + bias) + h_in (the order of k_linear's residual epilogue), then the next layer's LayerNorms with
  // k_ln_rows's arithmetic: its lane l holds features 8 l .. 8 l + 7 = group (t, u, h) here, l = 4 t + 2 u + h; its butterfly
  // folds lane bits 5 .. 0 = t bits 3 .. 0, u, h in that order.
  // Every row travels through the (now idle) ring as full 512-byte segments.  With the token on the lane a direct access touches 32 rows x 32 bytes per
  // instruction: the residual read, the h_out store and the two LayerNorm stores were ~320 such instructions per wave and took 131 k of a workgroup's 428 k clocks
  // (tools/bench_prefill.py --stamps, round 4) -- a third of the kernel.  Now a group of four tiles (128 features) of the wave's 32 tokens is one [32][132]-float
  // block of the wave's own LDS region: rows come in and go out 16 bytes per lane, two whole 512-byte segments per instruction; the token-on-lane view reads and
  // writes 32-byte pieces (row pitch 132 floats = 4 banks: eight consecutive rows tile the 32 banks, conflict-free ds_*_b128).  Same values, same order of operations.
  __syncthreads();                                      // every wave is done with the last weight chunk: the ring is staging space now
  constexpr int SROW = 132;                             // floats per staged row (128 + 4)
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SROW + 128);
  // the five parameter vectors of the epilogue (b_cat | g1 | b1 | g2 | b2, 512 floats each) behind the four staging blocks: 320 global loads per wave become LDS reads
  float* spar = reinterpret_cast<float*>(smem) + 4 * (32 * SROW + 128);
  for (int i = tid; i < 512; i += 256) {
    spar[i] = a.b_cat[i];
    if (a.nx1) { spar[512 + i] = a.g1[i]; spar[1024 + i] = a.b1[i]; spar[1536 + i] = a.g2[i]; spar[2048 + i] = a.b2[i]; }
  }
  __syncthreads();
  const int m0w = blockIdx.x * 128 + wave * 32;          // the wave's first token
  const int rrow = lane >> 5, rcol = lane & 31;          // row view: rows 2 it + rrow, 16-byte chunk rcol of the 512-byte segment
  f32x4 hv[16];
  auto load_rows = [&](int g) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      int mm = m0w + 2 * it + rrow; mm = mm < a.M ? mm : a.M - 1;
      hv[it] = *reinterpret_cast<const f32x4*>(a.hin + (long long)mm * 512 + g * 128 + rcol * 4);
    }
  };
  load_rows(0);
  float gs[16][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int it = 0; it < 16; ++it) *reinterpret_cast<f32x4*>(stg + (2 * it + rrow) * SROW + rcol * 4) = hv[it];
    if (g < 3) load_rows(g + 1);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 4 * g + tt, f0 = 32 * t + 16 * u + 8 * h;
        float* sp = stg + r * SROW + 32 * tt + 16 * u + 8 * h;
        const f32x4 ba = *reinterpret_cast<const f32x4*>(spar + f0), bb = *reinterpret_cast<const f32x4*>(spar + f0 + 4);
        const f32x4 ha = *reinterpret_cast<const f32x4*>(sp), hb = *reinterpret_cast<const f32x4*>(sp + 4);
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float va = (acc2[t][8 * u + j] + ba[j]) + ha[j], vb = (acc2[t][8 * u + 4 + j] + bb[j]) + hb[j];
          acc2[t][8 * u + j] = va; acc2[t][8 * u + 4 + j] = vb;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += acc2[t][8 * u + j];
        gs[t][u] = sm;
        const f32x4 oa = {acc2[t][8 * u], acc2[t][8 * u + 1], acc2[t][8 * u + 2], acc2[t][8 * u + 3]};
        const f32x4 ob = {acc2[t][8 * u + 4], acc2[t][8 * u + 5], acc2[t][8 * u + 6], acc2[t][8 * u + 7]};
        *reinterpret_cast<f32x4*>(sp) = oa;
        *reinterpret_cast<f32x4*>(sp + 4) = ob;
      }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int mm = m0w + 2 * it + rrow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (2 * it + rrow) * SROW + rcol * 4);
      if (mm < a.M) *reinterpret_cast<f32x4*>(a.hout + (long long)mm * 512 + g * 128 + rcol * 4) = v;
    }
  }
#ifdef ETD_PMLP_STAMP
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ps_t4 = __builtin_amdgcn_s_memtime();    // residual read + hout written
    if (lane == 0 && blockIdx.x < 64) {
      unsigned long long* o = g_pmlp_stamp + (blockIdx.x * 4 + wave) * 8;
      o[0] = ps_t2b - ps_t2; o[1] = ps_tx - ps_t0; o[2] = ps_acc[1] + ps_acc[2];
      o[3] = ps_t1 - ps_t0; o[4] = ps_t2 - ps_t1; o[5] = ps_t3 - ps_t2; o[6] = ps_t4 - ps_t3;
    }
  }
#endif
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
  // the two normalised rows, d16: halves of 256 features = 512-byte segments through the same staging block ([32][264] d16: the same 528-byte row pitch)
  d16* stb = reinterpret_cast<d16*>(stg);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const float* gam = spar + (which ? 1536 : 512); const float* bet = spar + (which ? 2048 : 1024);
    d16* dstp = which ? a.nx2 : a.nx1;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 8 * hh + tt, f0 = 32 * t + 16 * u + 8 * h;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + f0), gb = *reinterpret_cast<const f32x4*>(gam + f0 + 4);
          const f32x4 ba = *reinterpret_cast<const f32x4*>(bet + f0), bb = *reinterpret_cast<const f32x4*>(bet + f0 + 4);
          d16x8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = (d16)((acc2[t][8 * u + j] - mean) * rstd * ga[j] + ba[j]); o[4 + j] = (d16)((acc2[t][8 * u + 4 + j] - mean) * rstd * gb[j] + bb[j]);
          }
          *reinterpret_cast<d16x8*>(stb + r * (2 * SROW) + 32 * tt + 16 * u + 8 * h) = o;
        }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int mm = m0w + 2 * it + rrow;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stb + (2 * it + rrow) * (2 * SROW) + rcol * 8);
        if (mm < a.M) *reinterpret_cast<u32x4*>(dstp + (long long)mm * 512 + hh * 256 + rcol * 8) = v;
      }
    }
  }
#ifdef ETD_PMLP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && blockIdx.x < 64) g_pmlp_stamp[(blockIdx.x * 4 + wave) * 8 + 7] = __builtin_amdgcn_s_memtime() - ps_t0;       // the wave's whole life
#endif
}

#ifdef ETD_PMLP_STAMP
extern "C" int etd_debug_pmlp_stamps(unsigned long long* out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pmlp_stamp), sizeof(unsigned long long) * 64 * 4 * 8));
  return ETD_OK;
}
#endif
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

// Host side: dense_h_to_4h [2048][512] and the K-concatenated (dense_4h_to_h | attention.dense) [512][2560], both already d16 in
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
