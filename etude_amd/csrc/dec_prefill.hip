// Batched prefill of the EtudeDecoder, d16: ragged causal flash attention over every prompt of a begin_bars pass, reading K and V
// STRAIGHT FROM THE KV CACHE the QKV epilogue has just written                     modeling_gpt_neox.py:195-281, etude_decoder.py:291-297
//
// Rounds 1-3 ran the extractor's k_attn here, fed by two scratch copies the QKV epilogue wrote beside the cache rows: Kp (K again, row-major
// [row][hidden]) and a V^T image ([(sequence, head)][d][position], stored two bytes at a time from the token-on-lane accumulator) -- 1.5 KB of the
// 4.5 KB a prompt row cost in HBM writes, and a third of the QKV kernel's time.  The cache layout [slot][head][position][64] already IS what the
// kernel wants: a (sequence, head)'s keys are consecutive 128-byte rows (the K tile of k_attn with a row stride of 64), and V comes in the same
// shape, key-major.  The PV product needs V^T as the MFMA's A operand (d on the lane's row, keys along k): the V tile is staged key-major in LDS
// (192-byte rows: the four key rows of a transposing read then sit on four different 64-byte bank groups) and read with ds_read_b64_tr_b16, which
// hands every lane four consecutive keys of its d column.  Same products in the same order as k_attn on the scratch copies: bit-identical O.
//
// S^T = K Q^T with the QUERY on the lane (online-softmax state per lane), P^T feeds the PV MFMA from the accumulator registers.  Workgroup = 4 waves
// x 32 queries, key tiles of 64; the 128-query tiles are aligned to the END of the prompt (a prompt is 512 + 1 tokens after generate()'s truncation:
// right-aligned, the one-token tile is the first and visits one key tile).  A wave whose 32 queries all lie before a key tile skips the tile's
// arithmetic (its scores would all be masked: p = 0, alpha = 1 -- nothing changes but the sign of a zero).
#include "dec_kernels.h"
#include "prof.h"

#include <cstdlib>

#define PA_LDK 72     // K tile row stride (elements): 144 B, conflict-free ds_read_b128
#define PA_LDV 96     // V tile row stride (elements): 192 B -> key rows 4 h + q of a transposing read fall on bank groups 0, 192, 128, 64 (mod 256 B)

typedef __attribute__((ext_vector_type(4))) short pa_s16x4;
typedef __attribute__((address_space(3))) pa_s16x4* pa_lds_s16x4;

__global__ __launch_bounds__(256) void k_pattn(PAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * PA_LDK * 2 + 64 * PA_LDV * 2];
  d16* Ks = reinterpret_cast<d16*>(smem);
  d16* Vs = Ks + 64 * PA_LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int nh = a.n_heads;
  const int seq = blockIdx.y / nh, head = blockIdx.y - seq * nh;
  const int S = a.seq_len[seq];
  const long long r0 = a.seq_row0[seq];
  const int off = (128 - (S & 127)) & 127;                 // query tiles aligned to the end of the prompt
  if ((int)blockIdx.x * 128 - off >= S) return;            // whole workgroup beyond this (shorter) prompt
  const int slot = a.row_slot[r0];
  const int q0 = (int)blockIdx.x * 128 + wave * 32 - off;  // wave-uniform
  int qi = q0 + r; const bool qvalid = qi >= 0 && qi < S; qi = qi < 0 ? 0 : (qi < S ? qi : S - 1);

  const d16* qp = a.Q + (r0 + qi) * a.ldq + head * 64;
  d16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const d16x8*>(qp + s * 16 + h * 8);

  const long long cbase = (long long)slot * a.slot_stride + (long long)head * a.max_ctx * 64;
  const d16* kbase = a.Kc + cbase;
  const d16* vbase = a.Vc + cbase;

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  int ntile = (S + 63) >> 6;
  { const int lim = ((int)blockIdx.x * 128 + 127 - off) / 64 + 1; ntile = ntile < lim ? ntile : lim; }   // tiles past the workgroup's diagonal are fully masked
  // K / V tiles: 64 keys x 64 d each = 512 16-byte chunks, 2 per thread, one tile ahead in registers
  u32x4 kreg[2], vreg[2];
  auto tile_gload = [&](int kv0_) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      int key = kv0_ + row; key = key < S ? key : S - 1;       // rows past the prompt: a finite copy of the last key (their P is 0)
      kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (long long)key * 64 + ch * 8);
      vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (long long)key * 64 + ch * 8);
    }
  };
  // transposing read of the V tile: lane (16-lane group g, q = bits 3:2, p = bits 1:0) supplies the address of key row (4 h + q), columns 16 (g & 1) + 4 p .. + 4;
  // it receives keys 4 h .. 4 h + 3 of column d = lane & 31
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
  const d16* vtr = Vs + (4 * h + tq) * PA_LDV + 16 * tg + 4 * tp;
  tile_gload(0);
  for (int jt = 0; jt < ntile; ++jt) {
    const int kv0 = jt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(Ks + row * PA_LDK + ch * 8) = kreg[i];
      *reinterpret_cast<u32x4*>(Vs + row * PA_LDV + ch * 8) = vreg[i];
    }
    __syncthreads();
    if (jt + 1 < ntile) tile_gload(kv0 + 64);
    if (kv0 <= q0 + 31) {          // (wave-uniform) some query of this wave sees a key of this tile
      f32x16 sT[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sT[kt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const d16x8 kf = *reinterpret_cast<const d16x8*>(Ks + (kt * 32 + r) * PA_LDK + s * 16 + h * 8);
          sT[kt] = mfma32(kf, qf[s], sT[kt]);
        }
      }
      const bool need_mask = (kv0 + 64 > S) || (kv0 + 63 > q0);     // wave-uniform
      float mx = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kv0 + kt * 32 + acc_row(i, h);
            const float v = (key < S && key <= qi) ? sT[kt][i] : -INFINITY;
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
      const float c = a.scale_log2e;
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
      // O^T[d][query] += V^T[d][key] P^T[key][query]; k-step ks covers keys 16 ks .. + 15 in the accumulator's own order: element j <-> key 16 ks + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        d16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (d16)sT[ks >> 1][8 * (ks & 1) + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const pa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4)(vtr + (ks * 16) * PA_LDV + dt * 32));
          const pa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4)(vtr + (ks * 16 + 8) * PA_LDV + dt * 32));
          const __attribute__((ext_vector_type(8))) short v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = mfma32(__builtin_bit_cast(d16x8, v8), pf, o[dt]);
        }
      }
    }
    __syncthreads();
  }

  lrun += xhalf(lrun);
  const float inv = 1.f / lrun;
  if (qvalid) {
    d16* op = a.O + (r0 + q0 + r) * a.ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + 8 * q + 4 * h;
        *reinterpret_cast<d16x4*>(op + d) = pack4d(o[dt][4 * q] * inv, o[dt][4 * q + 1] * inv, o[dt][4 * q + 2] * inv, o[dt][4 * q + 3] * inv);
      }
  }
}

int launch_pattn(const PAttnArgs& a, hipStream_t st) {
  if (a.n_seq <= 0 || a.max_len <= 0 || a.n_heads <= 0 || !a.Q || !a.Kc || !a.Vc || !a.O || !a.seq_row0 || !a.seq_len || !a.row_slot || a.max_len > a.max_ctx ||
      (a.ldq % 8) || (a.ldo % 4) || (((uintptr_t)a.Q | (uintptr_t)a.Kc | (uintptr_t)a.Vc) & 15) || ((uintptr_t)a.O & 7))
    ETD_FAIL(ETD_EINVAL, "pattn: bad arguments");
  ETD_LAUNCH_FILTER("k_pattn");
  // bytes: Q read + O written once per row, K / V cache rows read once per (prompt, head) from HBM (the re-reads by later query tiles are L2 hits)
  ProfScope ps("k_pattn", st, a.flops_hint, ((double)a.n_seq * 4.0 * a.max_len * 64 * a.n_heads) * 2);
  dim3 g((a.max_len + 127) / 128, a.n_seq * a.n_heads);
  hipLaunchKernelGGL(k_pattn, g, dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_pqkv: the fused QKV projection of a batched prefill, X-STATIONARY              modeling_gpt_neox.py:195-207 (+ rotary :209-225, cache append)
//
// Rounds 1-3 ran this GEMM (M x 1536 x 512) on k_linear's 128 x 256 tile with the weight fragments loaded from global memory straight into operand registers:
// every wave pulls 4 KiB of fragments per 16-deep k-step through its CU's vector L1 -- eight waves ask for 128 B / clk of a 64 B / clk path, and the kernel ran at
// 635-780 TFLOP/s however its epilogue was arranged.  K = 512 is small enough to turn the loop inside out: a wave keeps its 32 tokens' whole input row block in
// registers (32 B-operand fragments = 128 registers, loaded once), the 1.5 MiB weight matrix streams through LDS in 32-feature tiles of 32 KiB (fragment order:
// an LDS-DMA piece is a plain copy, a fragment read a conflict-free ds_read_b128) shared by the workgroup's EIGHT waves -- 256 tokens per weight byte pulled from
// L2, two waves per SIMD -- and each tile is one chain of 32 MFMAs into 16 accumulator registers.  Two tiles are one (head, q | k | v) block of 64 features: bias,
// rotary embedding (lane-local: the pair (d, d + 8) sits in one lane) and d16 rounding on the token-on-lane accumulators, a transpose through the wave's own LDS
// block, and the rows leave as 128-byte segments: Q to the attention kernel's scratch, K / V to their cache rows -- 3 KB per prompt row, nothing else.
// The arithmetic is k_linear<QKV>'s (same MFMA, same k order, same epilogue formulas): bit-identical cache rows and queries (tools/bench_prefill.py --digest).
// Ring: two 32 KiB slots; tile t + 1 is requested at the top of tile t, behind the barrier that says every wave is done with tile t - 1's slot, and is waited
// for -- vmcnt(0), a whole tile of MFMAs later -- at the top of tile t + 1.  The row stores of a finished block are issued at the top of the NEXT tile, ahead
// of its 32 MFMAs, so that wait never waits for a store either.
// ================================================================================================
#define PQ_EPP 136                       // staging row stride (d16 elements): 272 B
#define PQ_TILE_ELEMS (32 * 512)         // one 32-feature tile of the fragment-ordered weights: 32 k-steps x 1 KiB
typedef const __attribute__((address_space(1))) void* pq_gptr_t;
typedef __attribute__((address_space(3))) void* pq_lptr_t;

// NW = waves per workgroup: 8 (256 tokens per weight byte pulled from L2; 141 KiB of LDS and 8 x 238 registers: the workgroup needs a whole, EMPTY CU) or 4 (128 tokens;
// 104 KiB and one wave per SIMD: half a CU's registers, so that a decode step's attention workgroups of another engine -- 8 waves x 88 registers, 19 KiB -- fit beside it and
// a CU that still holds such workgroups can take it: ETD_PQKV_WAVES=4, round 6's co-residency experiment)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_pqkv(PQkvArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PQ_TILE_ELEMS * 2 + NW * 32 * PQ_EPP * 2 + 1536 * 4];
  d16* ring = reinterpret_cast<d16*>(smem);
  d16* stg_all = reinterpret_cast<d16*>(smem + 2 * PQ_TILE_ELEMS * 2);
  float* sb = reinterpret_cast<float*>(smem + 2 * PQ_TILE_ELEMS * 2 + NW * 32 * PQ_EPP * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  d16* stg = stg_all + wave * (32 * PQ_EPP);
  const int NT = a.N >> 5;                                   // 48 tiles
  const int mw = blockIdx.x * (32 * NW) + wave * 32;         // the wave's first token
  const int m = mw + r, mc = m < a.M ? m : a.M - 1;

  // tile t -> slot t & 1: 32 one-KiB pieces, 32 / NW per wave
  auto issue = [&](int t) {
    constexpr int PW = 32 / NW;
    const d16* src = a.Wf + (long long)t * PQ_TILE_ELEMS + wave * (PW * 512) + lane * 8;
    d16* dst = ring + (t & 1) * PQ_TILE_ELEMS + wave * (PW * 512);
    // (one address, four immediate offsets: the offset field serves the global and the LDS address alike)
    __builtin_amdgcn_global_load_lds((pq_gptr_t)src, (pq_lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((pq_gptr_t)src, (pq_lptr_t)dst, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds((pq_gptr_t)src, (pq_lptr_t)dst, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds((pq_gptr_t)src, (pq_lptr_t)dst, 16, 3072, 0);
    if constexpr (PW == 8) {
      __builtin_amdgcn_global_load_lds((pq_gptr_t)(src + 2048), (pq_lptr_t)(dst + 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((pq_gptr_t)(src + 2048), (pq_lptr_t)(dst + 2048), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((pq_gptr_t)(src + 2048), (pq_lptr_t)(dst + 2048), 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((pq_gptr_t)(src + 2048), (pq_lptr_t)(dst + 2048), 16, 3072, 0);
    }
  };
  issue(0);
  for (int i = tid; i < a.N; i += 64 * NW) sb[i] = a.bias[i];
  // the wave's 32 tokens as B fragments: lane (token r, half h) holds x[token][16 s + 8 h .. + 8]
  d16x8 xf[32];
  {
    const d16* xp = a.X + (long long)mc * a.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 32; ++s) xf[s] = *reinterpret_cast<const d16x8*>(xp + 16 * s);
  }
  // rotary factors of this lane's token: dims 4 h .. 4 h + 3 (pair partner d + 8 = register i + 4)
  const int pos_l = a.rows.pos[mc];
  const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.rope_cos + (long long)pos_l * 8 + 4 * h);
  const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.rope_sin + (long long)pos_l * 8 + 4 * h);
  // the row-segment view of the wave's 32 x 64 block: lane -> row it * 8 + (lane >> 3), 16-byte chunk lane & 7
  const int er8 = lane >> 3, ec8 = lane & 7;
  long long qoff[4], coff[4];          // element offsets of the row's Q segment / cache row (without the head term); -1: nothing to store
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int mr = mw + it * 8 + er8, mrc = mr < a.M ? mr : a.M - 1;
    const int rpos = a.rows.pos[mrc], rslot = a.rows.slot[mrc], ract = a.rows.active[mrc];
    qoff[it] = mr < a.M ? (long long)mr * (a.n_heads * 64) + ec8 * 8 : -1;
    coff[it] = (mr < a.M && ract && rpos < a.max_ctx) ? (long long)rslot * a.slot_stride + (long long)rpos * 64 + ec8 * 8 : -1;
  }
  const unsigned ring_l = (unsigned)reinterpret_cast<uintptr_t>(ring) + lane * 16;

  // the rows of block u (features 64 u .. + 64: head u / 3, part u % 3) from the staging block to global memory
  auto store_block = [&](int u) {
    const int head = u / 3, part = u - head * 3;
    d16* cbase = (part == 1 ? a.Kc : a.Vc) + (long long)head * a.max_ctx * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const u32x4 val = *reinterpret_cast<const u32x4*>(stg + (it * 8 + er8) * PQ_EPP + ec8 * 8);
      if (part == 0) { if (qoff[it] >= 0) *reinterpret_cast<u32x4*>(a.Qb + qoff[it] + head * 64) = val; }
      else if (coff[it] >= 0) *reinterpret_cast<u32x4*>(cbase + coff[it]) = val;
    }
  };
  // ---- the tile's 32 chained MFMAs, hand-placed: hipcc reads two fragments, waits for them and multiplies twice (the chain then stalls on every pair of LDS reads); here
  // group g + 1's four fragment reads are in flight behind group g's four MFMAs (volatile asm in program order, counted lgkmcnt as in csrc/dec_fused.hip)
#define PQ_RD4(buf, F0)                                                                                                  \
  do {                                                                                                                   \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(buf[0]) : "v"(sl), "n"((F0) * 1024));                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(buf[1]) : "v"(sl), "n"((F0) * 1024 + 1024));                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(buf[2]) : "v"(sl), "n"((F0) * 1024 + 2048));                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(buf[3]) : "v"(sl), "n"((F0) * 1024 + 3072));                     \
  } while (0)
#define PQ_MFMA(c, af, bfr) asm volatile(ETD_MFMA16_DEC " %0, %1, %2, %0" : "+v"(c) : "v"(af), "v"(bfr))
#define PQ_MFMA0(c, af, bfr) asm volatile(ETD_MFMA16_DEC " %0, %1, %2, 0" : "=&v"(c) : "v"(af), "v"(bfr))
#define PQ_GROUP(c, G)                                                                                                   \
  do {                                                                                                                   \
    if constexpr ((G) < 7) { PQ_RD4(wf[((G) + 1) & 1], ((G) + 1) * 4); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); } \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    if constexpr ((G) == 0) PQ_MFMA0(c, wf[0][0], xf[0]); else PQ_MFMA(c, wf[(G) & 1][0], xf[4 * (G)]);                  \
    PQ_MFMA(c, wf[(G) & 1][1], xf[4 * (G) + 1]);                                                                         \
    PQ_MFMA(c, wf[(G) & 1][2], xf[4 * (G) + 2]);                                                                         \
    PQ_MFMA(c, wf[(G) & 1][3], xf[4 * (G) + 3]);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#define PQ_TILE(c, t)                                                                                                    \
  do {                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* this wave's pieces of tile t have landed (and the row stores of the block before) */ \
    __syncthreads();                                              /* ... everybody's; every wave is done reading tile t - 1's slot */ \
    if ((t) + 1 < NT) issue((t) + 1);                                                                                    \
    if ((t) >= 2 && !((t) & 1)) store_block(((t) - 2) >> 1);                                                              \
    const unsigned sl = ring_l + ((t) & 1) * (PQ_TILE_ELEMS * 2);                                                        \
    d16x8 wf[2][4];                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            /* (the staging reads of store_block are compiler-counted; start the hand count from zero) */ \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    PQ_RD4(wf[0], 0);                                                                                                    \
    PQ_GROUP(c, 0); PQ_GROUP(c, 1); PQ_GROUP(c, 2); PQ_GROUP(c, 3); PQ_GROUP(c, 4); PQ_GROUP(c, 5); PQ_GROUP(c, 6); PQ_GROUP(c, 7); \
  } while (0)
  for (int u = 0; u < (NT >> 1); ++u) {
    f32x16 c0, c1;
    PQ_TILE(c0, 2 * u);
    PQ_TILE(c1, 2 * u + 1);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(c1) : : "memory");      // (MFMA -> VALU wait states behind the asm MFMAs; c0's ended a whole tile ago)
    {
      // block u complete: bias, rotary embedding on Q / K, d16, transpose through the wave's staging block
      const int part = u % 3;
      float v[2][16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[0][i] = c0[i] + sb[64 * u + acc_row(i, h)]; v[1][i] = c1[i] + sb[64 * u + 32 + acc_row(i, h)]; }
      if (part < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x1 = v[0][i], x2 = v[0][i + 4];
          v[0][i] = x1 * c4[i] - x2 * s4[i];
          v[0][i + 4] = x2 * c4[i] + x1 * s4[i];
        }
      }
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<d16x4*>(stg + r * PQ_EPP + nn * 32 + 8 * q + 4 * h) = pack4d(v[nn][4 * q], v[nn][4 * q + 1], v[nn][4 * q + 2], v[nn][4 * q + 3]);
    }
  }
#undef PQ_TILE
#undef PQ_GROUP
#undef PQ_MFMA
#undef PQ_MFMA0
#undef PQ_RD4
  store_block((NT >> 1) - 1);
}

int launch_pqkv(const PQkvArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N != 1536 || a.ldx < 512 || (a.ldx % 8) || !a.X || !a.Wf || !a.bias || !a.Qb || !a.Kc || !a.Vc || !a.rows.pos || !a.rows.slot || !a.rows.active ||
      !a.rope_cos || !a.rope_sin || a.n_heads != 8 || (((uintptr_t)a.X | (uintptr_t)a.Wf | (uintptr_t)a.Qb | (uintptr_t)a.Kc | (uintptr_t)a.Vc) & 15) || (a.slot_stride % 8))
    ETD_FAIL(ETD_EINVAL, "pqkv: bad arguments");
  ETD_LAUNCH_FILTER("k_pqkv");
  ProfScope ps("k_pqkv", st, 2.0 * a.M * a.N * 512.0, ((double)a.M * 512 + (double)a.N * 512 + 3.0 * a.M * 512) * 2);
  static const int nw = getenv("ETD_PQKV_WAVES") ? atoi(getenv("ETD_PQKV_WAVES")) : 8;
  if (nw == 4) hipLaunchKernelGGL(k_pqkv<4>, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_pqkv<8>, dim3((a.M + 255) / 256), dim3(512), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
