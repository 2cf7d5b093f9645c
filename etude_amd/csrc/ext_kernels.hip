// hFT-Transformer (AMT-APC) kernels for gfx950: token-major e16 activations, fp32 accumulate,
// v_mfma_f32_32x32x16_bf16 everywhere.  Reference ops: etude/models/amt_apc.py (cited per kernel).
//
// Orientation convention ("swapped"): for Y = X W^T we issue mfma(A = W rows, B = X rows) so the
// accumulator holds Y^T: the TOKEN sits on the lane (col = lane&31) and the FEATURES sit in the 16
// registers (row = (i&3) + 8*(i>>2) + 4*(lane>>5)).  A token's row statistics (LayerNorm, softmax,
// argmax) are then lane-local plus one exchange with lane^32, and 4 consecutive features pack into
// one 8-byte row-major store.  Issuing the same two fragments the other way round gives Y with the
// feature on the lane, which is how V is written out pre-transposed (V^T) for the attention kernel.
#include "ext_kernels.h"
#include "prof.h"
#include "dec_epilogue.h"
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <vector>

#define LDK 72  // LDS row stride (elements) of a 64-wide e16 K-chunk: 144 B, 16-B aligned, conflict-free ds_read_b128
#define EPP 136 // epilogue staging row stride (e16 elements; 68 floats): 272 B = 17 x 16 B keeps rows 16-byte aligned, 128 features + pad

// ================================================================================================
// k_linear: Y = X W^T + b  (+ReLU | + residual + LayerNorm)       amt_apc.py:342-344,371,386-389,250,256
// Workgroup 256 threads = 4 waves on a 128-token x 256-feature tile.
// ================================================================================================
// Global -> register staging through buffer descriptors: a wave-uniform base (the workgroup's first X row / the weight
// block) in SGPRs, one 32-bit byte offset per lane and the K-chunk as scalar offset -- 12 address VGPRs instead of ~50 for
// flat 64-bit pointers (which is what lets the second X register set fit), and rows past M read as zeros (bounds check).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t lin_rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000);
}
// ---- loop A (extractor, K = 256 / 512): X and W both staged through LDS
__device__ __forceinline__ void lin_gload_x4(rsrc_t xs, const int (&xo)[4], int kc, u32x4 (&xr)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, xo[i], kc * 128, 0));
}
__device__ __forceinline__ void lin_gload_w8(rsrc_t ws, const int (&wo)[8], int kc, u32x4 (&wr)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) wr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, wo[i], kc * 128, 0));
}
__device__ __forceinline__ void lin_lstore_xw(e16* Xs, e16* Ws, int tid, const u32x4 (&xr)[4], const u32x4 (&wr)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    *reinterpret_cast<u32x4*>(Xs + row * LDK + ch * 8) = xr[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    *reinterpret_cast<u32x4*>(Ws + row * LDK + ch * 8) = wr[i];
  }
}

// (LDS-staged weights: the extractor's loop)  MODE 0: row-major store (+ReLU); MODE 1: V^T store (feature on the lane); MODE 2: residual + LayerNorm;
// MODE 10 + DEPI_x: EtudeDecoder epilogue x on the same tile (batched prefill of the Decode stage).
// Each of the 4 waves owns 64 tokens x 128 features (2 x 4 accumulator tiles): per 16-deep k-step it reads
// 2 X + 4 W fragments for 8 MFMAs (0.75 KB of LDS per MFMA; a 32 x 256 per-wave layout needs 1.125 KB and
// measured 2 % slower).
template <bool NORMAL_ORIENT>
__device__ __forceinline__ void lin_chunk_lds(const e16* Xs, const e16* Ws, int wm, int wn, int r, int h, f32x16 (&acc)[2][4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    e16x8 xf[2], wf[4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xf[mt] = *reinterpret_cast<const e16x8*>(Xs + (wm * 64 + mt * 32 + r) * LDK + s * 16 + h * 8);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = *reinterpret_cast<const e16x8*>(Ws + (wn * 128 + nt * 32 + r) * LDK + s * 16 + h * 8);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (NORMAL_ORIENT) acc[mt][nt] = mfma32(xf[mt], wf[nt], acc[mt][nt]);
        else               acc[mt][nt] = mfma32(wf[nt], xf[mt], acc[mt][nt]);
      }
  }
}

// ---- loop B (decoder prefill, K = 512 / 2560): W from fragment-ordered global memory straight into operand registers
__device__ __forceinline__ void lin_gload_x(rsrc_t xs, int xo, int row32_bytes, int kc, u32x4 (&xr)[4]) {   // rows (tid >> 3) + 32 i: per-lane offset + scalar offset
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, xo, kc * 128 + i * row32_bytes, 0));
}
__device__ __forceinline__ void lin_lstore_x(d16* Xs, int tid, const u32x4 (&xr)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256, row = c >> 3, ch = c & 7;
    *reinterpret_cast<u32x4*>(Xs + row * LDK + ch * 8) = xr[i];
  }
}
// Weights come in MFMA-FRAGMENT ORDER (pack_wfrag below / etd host loaders): block (n-tile t of 32 features, k-step s of 16)
// is 1 KiB, lane l holding row 32 t + (l & 31), columns 16 s + 8 (l >> 5) .. +8.  One load instruction per fragment reads
// one contiguous KiB straight into the operand registers: the weights never touch LDS (they were 2/3 of its traffic).
__device__ __forceinline__ d16x8 lin_wfrag(rsrc_t ws, int lane16, int blk) {
  return __builtin_bit_cast(d16x8, __builtin_amdgcn_raw_buffer_load_b128(ws, lane16, blk * 1024, 0));
}

// MODE 0: row-major store (+ReLU); MODE 1: V^T store (feature on the lane); MODE 2: residual + LayerNorm;
// MODE 10 + DEPI_x: EtudeDecoder epilogue x on the same tile (batched prefill of the Decode stage).
// Each of the 4 waves owns 64 tokens x 128 features (2 x 4 accumulator tiles): per 16-deep k-step it reads 2 X fragments
// from LDS and holds 4 W fragments in registers for 8 MFMAs.
// One 64-deep chunk: k-step s multiplies with wf[s][*], then re-requests those registers for the NEXT chunk's k-step s.
template <bool NORMAL_ORIENT>
__device__ __forceinline__ void lin_chunk(const d16* Xs, int wm, int r, int h, f32x16 (&acc)[2][4], d16x8 (&wf)[4][4],
                                          rsrc_t ws, int lane16, int wblk_next, int kblocks, bool more) {
  const d16* xp = Xs + (wm * 64 + r) * LDK + h * 8;
  d16x8 xf[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) xf[0][mt] = *reinterpret_cast<const d16x8*>(xp + mt * 32 * LDK);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    // the next k-step's X fragments are requested before this k-step's MFMAs (a k-step is 8 MFMAs = 256 clk; an LDS read
    // takes about that long to come back, and with the weights out of LDS nothing else hides it)
    if (s < 3) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xf[(s + 1) & 1][mt] = *reinterpret_cast<const d16x8*>(xp + mt * 32 * LDK + (s + 1) * 16);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (NORMAL_ORIENT) acc[mt][nt] = mfma32(xf[s & 1][mt], wf[s][nt], acc[mt][nt]);
        else               acc[mt][nt] = mfma32(wf[s][nt], xf[s & 1][mt], acc[mt][nt]);
      }
    if (more) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) wf[s][nt] = lin_wfrag(ws, lane16, wblk_next + nt * kblocks + s);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_linear(LinArgs a) {
  constexpr bool LN = MODE == 2;
  constexpr bool vt = MODE == 1;
  constexpr bool DEC = MODE >= 10;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(128 + 256) * LDK * 2 + 3 * 256 * 4];
  e16* Xs = reinterpret_cast<e16*>(smem);
  float* sb = reinterpret_cast<float*>(smem + (128 + 256) * LDK * 2);   // bias | gamma | beta
  float* lnred = reinterpret_cast<float*>(smem + 4 * 32 * EPP * 2);      // [2][128] after the K loop (LN mode), behind the epilogue staging tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroup -> tile: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The `nby`
  // feature blocks of one 128-token tile read the same X rows, so they take CONSECUTIVE slots of ONE XCD (id = 8*slot +
  // xcd, slot = tile_in_xcd * nby + block): the tile is fetched from HBM once and re-read from that XCD's L2.
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int mtile = (slot / a.nby) * 8 + xcd;
  if (mtile * 128 >= a.M) return;
  const int m0 = mtile * 128, nb = slot % a.nby + a.nb0, n0 = nb * 256, z = blockIdx.z;
  const e16* W = a.W + (long long)z * a.wz + (long long)n0 * a.K;
  const float* bias = a.bias + (long long)z * a.bz + n0;
  sb[tid] = bias[tid];
  if (LN) { sb[256 + tid] = a.gamma[tid]; sb[512 + tid] = a.beta[tid]; }

  f32x16 acc[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

#define LIN_STAMP(i) do { if (a.tbuf && tid == 0) a.tbuf[(long long)blockIdx.x * 16 + (i)] = clock64(); } while (0)
  // Two K loops, chosen per mode by measurement.  The decoder prefill (K = 512 and 2560, weights in fragment order) runs
  // loop B: -12 ... -18 % on its GEMMs.  The extractor (K = 256 / 512, two workgroups per CU) keeps loop A: with loop B both
  // token-halves of the workgroup fetch the same weight fragments through the CU's L1 and its windows took 5 % longer.
  if constexpr (DEC) {
  // K loop, 64-deep chunks.  X goes global -> registers -> LDS (two LDS buffers: one barrier per chunk); W goes global ->
  // operand registers, one chunk ahead.  Request order per chunk: W(c+1, s) behind k-step s, then X(c+3) at the end --
  // vmcnt retires in order, so the wait for a W fragment never forces the younger X requests.
  LIN_STAMP(0);
  u32x4 xa[4], xb[4];
  d16x8 wf[4][4];
  const int nk = (a.dbg & 2) ? 0 : a.K >> 6;       // even: K % 128 == 0 is checked by the launchers
  const int kblocks = a.K >> 4;                    // 16-deep k-steps per feature row block
  const rsrc_t xs = lin_rsrc(a.X + (long long)m0 * a.ldx, ((long long)(a.M - m0) * a.ldx) * 2);
  const rsrc_t ws = lin_rsrc(W, (long long)256 * a.K * 2);
  const int lane16 = lane * 16;
  const int wblk0 = __builtin_amdgcn_readfirstlane(wn) * 4 * kblocks;        // this wave's first fragment block (its 4 n-tiles are kblocks apart)
  const int xo = ((tid >> 3) * a.ldx + (tid & 7) * 8) * 2, xr32 = 32 * a.ldx * 2;
  d16* const Xd = reinterpret_cast<d16*>(smem);        // (the decoder modes compute in d16 whatever the extractor's element type is)
  d16* Xs1 = Xd + 128 * LDK;
  lin_gload_x(xs, xo, xr32, 0, xa);
#pragma unroll
  for (int sx = 0; sx < 4; ++sx)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[sx][nt] = lin_wfrag(ws, lane16, wblk0 + nt * kblocks + sx);
  lin_gload_x(xs, xo, xr32, nk > 1 ? 1 : 0, xb);
  __builtin_amdgcn_sched_barrier(0);
  lin_lstore_x(Xd, tid, xa);
  if (nk > 2) lin_gload_x(xs, xo, xr32, 2, xa);
  __syncthreads();
  LIN_STAMP(1);
  for (int kc = 0; kc < nk; kc += 2) {
    lin_chunk<vt>(Xd, wm, r, h, acc, wf, ws, lane16, wblk0 + (kc + 1) * 4, kblocks, true);       // kc + 1 < nk always (nk even)
    lin_lstore_x(Xs1, tid, xb);
    if (kc + 3 < nk) lin_gload_x(xs, xo, xr32, kc + 3, xb);
    __syncthreads();
    if (kc == 0) LIN_STAMP(2);
    lin_chunk<vt>(Xs1, wm, r, h, acc, wf, ws, lane16, wblk0 + (kc + 2) * 4, kblocks, kc + 2 < nk);
    if (kc + 2 < nk) {
      lin_lstore_x(Xd, tid, xa);
      if (kc + 4 < nk) lin_gload_x(xs, xo, xr32, kc + 4, xa);
    }
    __syncthreads();
    if (kc == 0) { LIN_STAMP(3); LIN_STAMP(4); }
  }
  } else {
    e16* Ws = Xs + 128 * LDK;
  // K loop, 64-deep chunks through one LDS buffer.  The activation rows come from HBM / the Infinity Cache with ~2-3 us of
  // latency under load while a chunk's 32 MFMAs per wave take 0.4 us, so X is requested TWO chunks ahead (two register
  // sets, alternating); the weight chunk (L2-resident) one ahead, and before the X request of the same step so that the
  // in-order vmcnt wait for it leaves the younger X loads in flight.
  LIN_STAMP(0);
  u32x4 xa[4], xb[4], wr[8];
  const int nk = (a.dbg & 2) ? 0 : a.K >> 6;       // even: K % 128 == 0 is checked by the launchers
  const rsrc_t xs = lin_rsrc(a.X + (long long)m0 * a.ldx, ((long long)(a.M - m0) * a.ldx) * 2);
  const rsrc_t ws = lin_rsrc(W, (long long)256 * a.K * 2);
  int xo[4], wo[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = tid + i * 256; xo[i] = ((c >> 3) * a.ldx + (c & 7) * 8) * 2; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int c = tid + i * 256; wo[i] = ((c >> 3) * a.K + (c & 7) * 8) * 2; }
  lin_gload_x4(xs, xo, 0, xa);
  lin_gload_w8(ws, wo, 0, wr);
  lin_gload_x4(xs, xo, nk > 1 ? 1 : 0, xb);
  for (int kc = 0; kc < nk; kc += 2) {
    lin_lstore_xw(Xs, Ws, tid, xa, wr);
    __syncthreads();
    if (kc == 0) LIN_STAMP(1);
    lin_gload_w8(ws, wo, kc + 1, wr);                                        // kc + 1 < nk always (nk even)
    if (kc + 2 < nk) lin_gload_x4(xs, xo, kc + 2, xa);
    lin_chunk_lds<vt>(Xs, Ws, wm, wn, r, h, acc);
    __syncthreads();
    if (kc == 0) LIN_STAMP(2);
    lin_lstore_xw(Xs, Ws, tid, xb, wr);
    __syncthreads();
    if (kc == 0) LIN_STAMP(3);
    if (kc + 2 < nk) lin_gload_w8(ws, wo, kc + 2, wr);
    if (kc + 3 < nk) lin_gload_x4(xs, xo, kc + 3, xb);
    lin_chunk_lds<vt>(Xs, Ws, wm, wn, r, h, acc);
    __syncthreads();
    if (kc == 0) LIN_STAMP(4);
  }
  }
  LIN_STAMP(5);

  if ((a.dbg & 1) && acc[0][0][0] != 1.2345e30f && acc[1][3][15] != 1.2345e30f) return;

  // ---- epilogue.  The accumulator has the TOKEN on the lane, so a direct store touches 64 different rows with 8 bytes
  // each; the memory pipeline then handles one row segment at a time and the store phase costs as much as the K loop
  // (measured at K = 256: 53 % of the kernel).  Instead every wave transposes its 32-token x 128-feature half tile through
  // a private LDS region (the K-loop buffers are free now) and reads / writes global memory in full row segments, 16 bytes
  // per lane: 4 rows x 256 B per instruction.
  e16* stg = reinterpret_cast<e16*>(smem) + wave * (32 * EPP);
  float* stf = reinterpret_cast<float*>(smem) + wave * (32 * (EPP / 2));
  const int er = lane >> 4, ec = lane & 15;          // row-contiguous view: row = 4*it + er, 16-byte chunk ec
  const int mw = m0 + wm * 64, nw = n0 + wn * 128;   // first token / feature of this wave's tile

  if constexpr (DEC) {
    constexpr int EPI = MODE - 10;
    const DGemmArgs& g = a.dec;
    if constexpr (EPI == DEPI_GELU) {
      // MLP up: bias + erf-GELU, bf16 rows for the down projection          modeling_gpt_neox.py:239-245
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int fl = nt * 32 + 8 * q + 4 * h, f = wn * 128 + fl;
            *reinterpret_cast<d16x4*>(stg + r * EPP + fl) =
                pack4d(gelu_fast(acc[mt][nt][4 * q] + sb[f]), gelu_fast(acc[mt][nt][4 * q + 1] + sb[f + 1]),
                      gelu_fast(acc[mt][nt][4 * q + 2] + sb[f + 2]), gelu_fast(acc[mt][nt][4 * q + 3] + sb[f + 3]));
          }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + er, m = mw + mt * 32 + row;
          if (m < a.M) *reinterpret_cast<u32x4*>(g.Yb + (long long)m * g.ldy + nw + ec * 8) = *reinterpret_cast<const u32x4*>(stg + row * EPP + ec * 8);
        }
        __syncthreads();
      }
    } else if constexpr (EPI == DEPI_RESID) {
      // parallel residual: hout = ((acc + bias) [+ add]) + hin, all fp32      modeling_gpt_neox.py:272
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ntp = 0; ntp < 2; ++ntp) {
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int nt = ntp * 2 + nn, fl = nn * 32 + 8 * q + 4 * h, f = wn * 128 + ntp * 64 + fl;
              const f32x4 o = {acc[mt][nt][4 * q] + sb[f], acc[mt][nt][4 * q + 1] + sb[f + 1], acc[mt][nt][4 * q + 2] + sb[f + 2], acc[mt][nt][4 * q + 3] + sb[f + 3]};
              *reinterpret_cast<f32x4*>(stf + r * (EPP / 2) + fl) = o;
            }
          __syncthreads();
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + er, m = mw + mt * 32 + row;
            if (m < a.M) {
              const long long off = (long long)m * g.N + nw + ntp * 64 + ec * 4;
              f32x4 v = *reinterpret_cast<const f32x4*>(stf + row * (EPP / 2) + ec * 4);
              if (g.add) { const f32x4 ad = *reinterpret_cast<const f32x4*>(g.add + off); v[0] += ad[0]; v[1] += ad[1]; v[2] += ad[2]; v[3] += ad[3]; }
              const f32x4 hi = *reinterpret_cast<const f32x4*>(g.hin + off);
              const f32x4 o = {v[0] + hi[0], v[1] + hi[1], v[2] + hi[2], v[3] + hi[3]};
              *reinterpret_cast<f32x4*>(g.hout + off) = o;
            }
          }
          __syncthreads();
        }
    } else if constexpr (EPI == DEPI_QKV) {
      if (!g.Qb) {
        // (no MFMA-attention scratch: the prompt is longer than the scratch rows) -- token-on-lane epilogue
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = mw + mt * 32 + r;
          if (m < a.M) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) dgemm_epilogue<true, EPI>(g, acc[mt][nt], m, nw + nt * 32, h);
          }
        }
      } else {
        // fused QKV laid out [head][q|k|v][64] (modeling_gpt_neox.py:204-207): a 64-feature pair of accumulator tiles is one
        // (head, part).  RoPE on the token-on-lane values, then the Q rows and the KV-cache rows leave as 128-byte segments: 3 KB per
        // prompt row (the prefill attention reads K / V from those cache rows, csrc/dec_prefill.hip).
        const int er8 = lane >> 3, ec8 = lane & 7;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int ml = mw + mt * 32 + r, mlc = ml < a.M ? ml : a.M - 1;
          const int pos_l = g.rows.pos[mlc];
          int rpos[4], rslot[4], ract[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            int m = mw + mt * 32 + it * 8 + er8; m = m < a.M ? m : a.M - 1;
            rpos[it] = g.rows.pos[m]; rslot[it] = g.rows.slot[m]; ract[it] = g.rows.active[m];
          }
#pragma unroll
          for (int ntp = 0; ntp < 2; ++ntp) {
            const int fb = nw + ntp * 64, head = fb / 192, part = (fb - head * 192) >> 6;
            float v[2][16];
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
              for (int i = 0; i < 16; ++i) v[nn][i] = acc[mt][ntp * 2 + nn][i] + sb[wn * 128 + ntp * 64 + nn * 32 + acc_row(i, h)];
            if (part < 2) {
              // partial RoPE on dims [0, 16): pair (d, d + 8); d = (i&3) + 4h (i < 4), d + 8 = register i + 4 of the same lane
              const f32x4 c4 = *reinterpret_cast<const f32x4*>(g.rope_cos + (long long)pos_l * 8 + 4 * h);
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(g.rope_sin + (long long)pos_l * 8 + 4 * h);
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
                *reinterpret_cast<d16x4*>(stg + r * EPP + nn * 32 + 8 * q + 4 * h) = pack4d(v[nn][4 * q], v[nn][4 * q + 1], v[nn][4 * q + 2], v[nn][4 * q + 3]);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int row = it * 8 + er8, m = mw + mt * 32 + row;
              if (m < a.M) {
                const u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * EPP + ec8 * 8);
                if (part == 0) *reinterpret_cast<u32x4*>(g.Qb + (long long)m * (g.n_heads * 64) + head * 64 + ec8 * 8) = val;
                else {
                  if (ract[it] && rpos[it] < g.max_ctx) {
                    d16* cp = reinterpret_cast<d16*>(part == 1 ? g.Kc : g.Vc) + (long long)rslot[it] * g.slot_stride + ((long long)head * g.max_ctx + rpos[it]) * 64 + ec8 * 8;
                    *reinterpret_cast<u32x4*>(cp) = val;
                  }
                }
              }
            }
            __syncthreads();
          }
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = mw + mt * 32 + r;
        if (m < a.M) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) dgemm_epilogue<true, EPI>(g, acc[mt][nt], m, nw + nt * 32, h);
        }
      }
    }
  } else if constexpr (!LN) {
    if constexpr (vt) {
      // V^T rows: with the feature on the lane a direct store puts 8 bytes into each of 32 different rows per instruction.  When a
      // wave's 64 tokens are consecutive positions of ONE sequence (S % 64 == 0), the 32-feature x 64-token sub-tile goes through the
      // wave's LDS region instead and leaves as full 128-byte row segments (8 rows per instruction).
      const bool rowseg = (a.S % 64 == 0) && (a.M % 64 == 0) && (a.Spad % 8 == 0) && (a.vtz % 8 == 0) && ((reinterpret_cast<uintptr_t>(a.VT) & 15) == 0) && !(a.dbg & 4);
      if (rowseg) {
        const int mb = m0 + wm * 64, seq = mb / a.S, pos0 = mb - seq * a.S;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float b = sb[wn * 128 + nt * 32 + r];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<e16x4*>(stg + r * 72 + mt * 32 + 8 * q + 4 * h) =
                  pack4e(acc[mt][nt][4 * q] + b, acc[mt][nt][4 * q + 1] + b, acc[mt][nt][4 * q + 2] + b, acc[mt][nt][4 * q + 3] + b);
          __syncthreads();
          if (mb < a.M) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int row = it * 8 + (lane >> 3), ch = lane & 7, fr = wn * 128 + nt * 32 + row;
              e16* dst = a.VT + (long long)z * a.vtz + ((long long)(seq * 4 + (fr >> 6)) * 64 + (fr & 63)) * a.Spad + pos0 + ch * 8;
              *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(stg + row * 72 + ch * 8);
            }
          }
          __syncthreads();
        }
      } else
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int f = wn * 128 + nt * 32 + r;
        const float b = sb[f];
        const int head = f >> 6, d = f & 63;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm * 64 + mt * 32 + 8 * q + 4 * h;
            if (m < a.M) {
              const int seq = m / a.S, pos = m - seq * a.S;
              e16* dst = a.VT + (long long)z * a.vtz + ((long long)(seq * 4 + head) * 64 + d) * a.Spad + pos;
              *reinterpret_cast<e16x4*>(dst) = pack4e(acc[mt][nt][4 * q] + b, acc[mt][nt][4 * q + 1] + b, acc[mt][nt][4 * q + 2] + b, acc[mt][nt][4 * q + 3] + b);
            }
          }
      }
    } else {
      e16* ybase = a.Y + (long long)z * a.yz + nw + ec * 8;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int fl = nt * 32 + 8 * q + 4 * h, f = wn * 128 + fl;
            float v0 = acc[mt][nt][4 * q + 0] + sb[f + 0], v1 = acc[mt][nt][4 * q + 1] + sb[f + 1];
            float v2 = acc[mt][nt][4 * q + 2] + sb[f + 2], v3 = acc[mt][nt][4 * q + 3] + sb[f + 3];
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            *reinterpret_cast<e16x4*>(stg + r * EPP + fl) = pack4e(v0, v1, v2, v3);
          }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + er, m = mw + mt * 32 + row;
          if (m < a.M) *reinterpret_cast<u32x4*>(ybase + (long long)m * a.ldy) = *reinterpret_cast<const u32x4*>(stg + row * EPP + ec * 8);
        }
        __syncthreads();
        LIN_STAMP(6 + mt);
      }
    }
  } else {
    // residual + LayerNorm: a token's 256 features live in 2 waves (wn = 0, 1) x 2 half-lanes; the halves combine
    // with one lane exchange, the waves through LDS in a fixed order.  The residual rows come in through the same
    // staging region (row-segment loads), the normalised rows leave through it.
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + er, m = mw + mt * 32 + row;
        const int mc = m < a.M ? m : a.M - 1;
        const int rrow = a.r_mod > 0 ? mc % a.r_mod : mc;
        *reinterpret_cast<u32x4*>(stg + row * EPP + ec * 8) = *reinterpret_cast<const u32x4*>(a.R + (long long)rrow * a.ldr + wn * 128 + ec * 8);
      }
      __syncthreads();
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int fl = nt * 32 + 8 * q + 4 * h, f = wn * 128 + fl;
          const e16x4 rv = *reinterpret_cast<const e16x4*>(stg + r * EPP + fl);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[mt][nt][4 * q + j] + sb[f + j] + bf2f(rv[j]);
            acc[mt][nt][4 * q + j] = v;
            s += v;
          }
        }
      s += xhalf(s);
      if (h == 0) lnred[wn * 128 + wm * 64 + mt * 32 + r] = s;
      __syncthreads();
    }
    float mean[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int tl = wm * 64 + mt * 32 + r;
      mean[mt] = (lnred[tl] + lnred[128 + tl]) * (1.f / 256.f);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float dlt = acc[mt][nt][i] - mean[mt]; s += dlt * dlt; }
      s += xhalf(s);
      if (h == 0) lnred[wn * 128 + wm * 64 + mt * 32 + r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int tl = wm * 64 + mt * 32 + r;
      const float rstd = rsqrtf((lnred[tl] + lnred[128 + tl]) * (1.f / 256.f) + 1e-5f);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int fl = nt * 32 + 8 * q + 4 * h, f = wn * 128 + fl;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (acc[mt][nt][4 * q + j] - mean[mt]) * rstd * sb[256 + f + j] + sb[512 + f + j];
          *reinterpret_cast<e16x4*>(stg + r * EPP + fl) = pack4e(v[0], v[1], v[2], v[3]);
        }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + er, m = mw + mt * 32 + row;
        if (m < a.M) *reinterpret_cast<u32x4*>(a.Y + (long long)m * a.ldy + wn * 128 + ec * 8) = *reinterpret_cast<const u32x4*>(stg + row * EPP + ec * 8);
      }
      __syncthreads();
    }
  }
}

// token tiles padded to a multiple of 8 (one per XCD), times the feature blocks of the launch
static unsigned lin_grid_x(int M, int nby) { return (unsigned)(((M + 127) / 128 + 7) / 8 * 8 * nby); }
static int lin_dbg() { static const int v = ETD_XENV("ETD_LIN_DBG") ? atoi(ETD_XENV("ETD_LIN_DBG")) : 0; return v; }
int launch_linear(const LinArgs& a, int nz, hipStream_t st) {
  if (a.K % 128 || a.N % 256 || a.M <= 0) ETD_FAIL(ETD_EINVAL, "linear: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.vt_block >= 0 && (a.S % 4 || a.Spad % 4 || !a.VT)) ETD_FAIL(ETD_EINVAL, "linear: bad V^T args");
  if (a.Y && (a.ldy % 8 || a.yz % 8 || ((uintptr_t)a.Y & 15))) ETD_FAIL(ETD_EINVAL, "linear: Y rows must be 16-byte aligned (ldy=%d)", a.ldy);
  // row-major blocks [0, vt_block) (or all), then the V^T block as its own launch (orientation is a
  // compile-time property of the MFMA loop)
  ETD_LAUNCH_FILTER("k_linear");
  ProfScope ps("k_linear", st, 2.0 * a.M * a.N * a.K * nz, ((double)a.M * a.K + (double)a.N * a.K * nz + (double)a.M * a.N * nz) * 2);
  const int nblk = a.N / 256;
  const int n_plain = a.vt_block >= 0 ? a.vt_block : nblk;
  if (a.vt_block >= 0 && a.vt_block != nblk - 1) ETD_FAIL(ETD_EINVAL, "linear: V^T block must be the last block");
  if (n_plain > 0) {
    LinArgs b = a; b.nb0 = 0; b.dbg = lin_dbg(); b.nby = n_plain;
    static int stamp_left = getenv("ETD_LIN_STAMP") ? atoi(getenv("ETD_LIN_STAMP")) : 0;
    const unsigned gx = lin_grid_x(a.M, n_plain);
    if (stamp_left > 0 && nz == 1) {
      --stamp_left;
      long long* tb = nullptr;
      (void)hipMalloc(&tb, (size_t)gx * 16 * 8);
      (void)hipMemsetAsync(tb, 0, (size_t)gx * 16 * 8, st);
      b.tbuf = tb;
      hipLaunchKernelGGL(k_linear<0>, dim3(gx, 1, nz), dim3(256), 0, st, b);
      (void)hipStreamSynchronize(st);
      std::vector<long long> hb((size_t)gx * 16);
      (void)hipMemcpy(hb.data(), tb, hb.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(tb);
      double d[8] = {0}; long long n = 0, tmin = -1, tmax = 0;
      for (unsigned g = 0; g < gx; ++g) {
        const long long* t = &hb[(size_t)g * 16];
        if (!t[0] || !t[7]) continue;
        ++n;
        for (int i = 0; i < 7; ++i) d[i] += (double)(t[i + 1] - t[i]);
        if (tmin < 0 || t[0] < tmin) tmin = t[0];
        if (t[7] > tmax) tmax = t[7];
      }
      fprintf(stderr, "[lin stamp] M=%d N=%d K=%d relu=%d wgs=%lld span=%lld clk | load0+lstore %.0f | chunk0 %.0f | lstore1 %.0f | chunk1 %.0f | rest of K loop %.0f | epi0 %.0f | epi1 %.0f\n",
              a.M, a.N, a.K, a.relu, n, tmax - tmin, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n);
      return ETD_OK;
    }
    hipLaunchKernelGGL(k_linear<0>, dim3(gx, 1, nz), dim3(256), 0, st, b);
  }
  if (a.vt_block >= 0) {
    LinArgs b = a; b.nb0 = a.vt_block; b.dbg = lin_dbg(); b.nby = 1;
    hipLaunchKernelGGL(k_linear<1>, dim3(lin_grid_x(a.M, 1), 1, nz), dim3(256), 0, st, b);
  }
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
int launch_linear_dec(const LinArgs& a, int dec_epi, hipStream_t st) {
  if (a.K % 128 || a.N % 256 || a.M <= 0 || !a.bias) ETD_FAIL(ETD_EINVAL, "linear_dec: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (dec_epi == DEPI_GELU && (!a.dec.Yb || a.dec.ldy % 8 || ((uintptr_t)a.dec.Yb & 15))) ETD_FAIL(ETD_EINVAL, "linear_dec: GELU needs 16-byte aligned d16 rows");
  if (dec_epi == DEPI_RESID && (!a.dec.hin || !a.dec.hout || a.dec.N % 4)) ETD_FAIL(ETD_EINVAL, "linear_dec: bad residual arguments");
  if (dec_epi == DEPI_QKV && a.dec.Qb && (a.dec.rot_half != 8 || a.N % 192)) ETD_FAIL(ETD_EINVAL, "linear_dec: bad QKV arguments");
  ETD_LAUNCH_FILTER("k_linear_dec");
  ProfScope ps("k_linear_dec", st, 2.0 * a.M * a.N * a.K, ((double)a.M * a.K + (double)a.N * a.K) * 2);
  dim3 g(lin_grid_x(a.M, a.N / 256), 1, 1);
  LinArgs b = a; b.nb0 = 0; b.dbg = lin_dbg(); b.nby = a.N / 256;
  switch (dec_epi) {
    case DEPI_BIAS: hipLaunchKernelGGL(k_linear<10 + DEPI_BIAS>, g, dim3(256), 0, st, b); break;
    case DEPI_GELU: hipLaunchKernelGGL(k_linear<10 + DEPI_GELU>, g, dim3(256), 0, st, b); break;
    case DEPI_RESID: hipLaunchKernelGGL(k_linear<10 + DEPI_RESID>, g, dim3(256), 0, st, b); break;
    case DEPI_QKV: hipLaunchKernelGGL(k_linear<10 + DEPI_QKV>, g, dim3(256), 0, st, b); break;
    default: ETD_FAIL(ETD_EINVAL, "linear_dec: unsupported epilogue %d", dec_epi);
  }
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

int launch_linear_ln(const LinArgs& a, hipStream_t st) {
  if (a.K % 128 || a.N != 256 || a.M <= 0 || !a.R || !a.gamma || !a.beta || a.ldr % 8 || a.ldy % 8 || (((uintptr_t)a.R | (uintptr_t)a.Y) & 15))
    ETD_FAIL(ETD_EINVAL, "linear_ln: bad args");
  ETD_LAUNCH_FILTER("k_linear_ln");
  ProfScope ps("k_linear_ln", st, 2.0 * a.M * a.N * a.K, ((double)a.M * a.K + (double)a.N * a.K + 2.0 * a.M * a.N) * 2);
  dim3 g(lin_grid_x(a.M, 1), 1, 1);
  LinArgs b = a; b.nb0 = 0; b.dbg = lin_dbg(); b.nby = 1;
  hipLaunchKernelGGL(k_linear<2>, g, dim3(256), 0, st, b);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_attn: softmax(Q K^T / 8) V per (sequence, head)                           amt_apc.py:349-368
// S^T = K Q^T is accumulated with the QUERY on the lane, so the online-softmax state (m, l) and the
// rescale of O^T are per-lane scalars; P^T feeds the PV product straight from the accumulator
// registers (guide §3 "accumulator tile as the next MFMA's operand"), V^T comes from LDS.
// Workgroup = 4 waves x 32 queries; KV tiles of 64 keys.
// ================================================================================================
#define LDV 68  // V^T tile row stride (elements): 136 B -> conflict-free ds_read_b64 across 32 d-rows

__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * LDK * 2 + 64 * LDV * 2];
  e16* Ks = reinterpret_cast<e16*>(smem);
  e16* Vs = Ks + 64 * LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int nh = a.n_heads > 0 ? a.n_heads : 4;
  const int seq = blockIdx.y / nh, head = blockIdx.y - seq * nh;
  int Sq = a.Sq, Sk = a.Sk;
  long long qbase = (long long)seq * a.q_seq_stride, kbase_off = (long long)seq * a.k_seq_stride, obase = (long long)seq * a.o_seq_stride;
  // Ragged causal batches (the decoder's batched prefill): the 128-query tiles are aligned to the END of the sequence.  A prompt is 512 + 1 tokens after
  // generate()'s truncation; left-aligned, its 513th token gets a tile of its own that walks all 9 key tiles (29 tile-steps per (sequence, head)); right-aligned the
  // ragged tile is the FIRST, with one key tile to visit (25).  Per query the same keys in the same order: bit-identical.
  int off = 0;
  if (a.seq_len) {                                   // ragged batch: rows of this sequence
    Sq = Sk = a.seq_len[seq];
    const long long r0 = a.seq_row0[seq];
    qbase = r0 * a.ldq; kbase_off = r0 * a.ldk; obase = r0 * a.ldo;
    if (a.causal) off = (128 - (Sq & 127)) & 127;
    if ((int)blockIdx.x * 128 - off >= Sq) return;   // whole workgroup beyond this (shorter) sequence
  }
  const int q0 = (int)blockIdx.x * 128 + wave * 32 - off;
  int qi = q0 + r; const bool qvalid = qi >= 0 && qi < Sq; qi = qi < 0 ? 0 : (qi < Sq ? qi : Sq - 1);

  const e16* qp = a.Q + qbase + (long long)qi * a.ldq + head * 64;
  e16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const e16x8*>(qp + s * 16 + h * 8);

  const e16* kbase = a.K + kbase_off + head * 64;
  const e16* vbase = a.VT + ((long long)(seq * nh + head) * 64) * a.Spad;

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  int ntile = (Sk + 63) >> 6;
  if (a.causal) { const int lim = ((int)blockIdx.x * 128 + 127 - off) / 64 + 1; ntile = ntile < lim ? ntile : lim; }   // tiles past the diagonal are fully masked
  // K/V tiles are prefetched one tile ahead into registers (global latency hides under the previous tile's MFMA/softmax)
  u32x4 kreg[2], vreg[2];
  auto tile_gload = [&](int kv0_) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      int key = kv0_ + row; key = key < Sk ? key : Sk - 1;
      kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (long long)key * a.ldk + ch * 8);
      vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (long long)row * a.Spad + kv0_ + ch * 8);
    }
  };
  tile_gload(0);
  for (int jt = 0; jt < ntile; ++jt) {
    const int kv0 = jt * 64;
    // stage K tile [64 keys][64 d] and V^T tile [64 d][64 keys]: 512 16-B chunks each, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(Ks + row * LDK + ch * 8) = kreg[i];
      u32x2* vd = reinterpret_cast<u32x2*>(Vs + row * LDV + ch * 8);
      const u32x2 lo = {vreg[i][0], vreg[i][1]}, hi = {vreg[i][2], vreg[i][3]};
      vd[0] = lo;
      vd[1] = hi;
    }
    __syncthreads();
    if (jt + 1 < ntile) tile_gload(kv0 + 64);

    // S^T[key][query] for the two 32-key sub-tiles
    f32x16 sT[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sT[kt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const e16x8 kf = *reinterpret_cast<const e16x8*>(Ks + (kt * 32 + r) * LDK + s * 16 + h * 8);
        sT[kt] = mfma32(kf, qf[s], sT[kt]);
      }
    }
    // mask (only tiles that need it), running max of the RAW scores; the softmax scale rides in the exp2 argument:
    // p = exp2(s*c - m*c) is one fma + one v_exp_f32 per element (this loop is VALU-bound: 256 FLOP per exp at d = 64)
    const bool need_mask = (kv0 + 64 > Sk) || (a.causal && kv0 + 63 > q0);     // wave-uniform
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kv0 + kt * 32 + acc_row(i, h);
          const float v = (key < Sk && (!a.causal || key <= qi)) ? sT[kt][i] : -INFINITY;
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
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);   // raw v_exp_f32: argument <= 0
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

    // O^T[d][query] += V^T[d][key] * P^T[key][query]; k-step ks covers keys 16ks..16ks+15 in the
    // accumulator's own (permuted) order: element j <-> key 16ks + 8(j>>2) + 4h + (j&3)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      e16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (e16)sT[ks >> 1][8 * (ks & 1) + j];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const e16* vrow = Vs + (dt * 32 + r) * LDV + ks * 16 + 4 * h;
        const e16x4 lo = *reinterpret_cast<const e16x4*>(vrow);
        const e16x4 hi = *reinterpret_cast<const e16x4*>(vrow + 8);
        const e16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = mfma32(vf, pf, o[dt]);
      }
    }
    __syncthreads();
  }

  lrun += xhalf(lrun);
  const float inv = 1.f / lrun;
  if (qvalid) {
    e16* op = a.O + obase + (long long)(q0 + r) * a.ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + 8 * q + 4 * h;
        *reinterpret_cast<e16x4*>(op + d) = pack4e(o[dt][4 * q] * inv, o[dt][4 * q + 1] * inv, o[dt][4 * q + 2] * inv, o[dt][4 * q + 3] * inv);
      }
  }
}

int launch_attn(const AttnArgs& a, hipStream_t st) {
  if (a.Sq <= 0 || a.Sk <= 0 || a.n_seq <= 0 || a.Spad % 64 || a.Spad < ((a.Sk + 63) / 64) * 64)
    ETD_FAIL(ETD_EINVAL, "attn: bad shape Sq=%d Sk=%d Spad=%d", a.Sq, a.Sk, a.Spad);
  const int nh = a.n_heads > 0 ? a.n_heads : 4;
  ETD_LAUNCH_FILTER(a.causal ? "k_attn_causal" : "k_attn");
  ProfScope ps(a.causal ? "k_attn_causal" : "k_attn", st, a.flops_hint > 0 ? a.flops_hint : (a.causal ? 0.5 : 1.0) * 256.0 * nh * a.n_seq * a.Sq * a.Sk, ((double)a.n_seq * (2.0 * a.Sq + 2.0 * a.Sk) * 64 * nh) * 2);
  dim3 g((a.Sq + 127) / 128, a.n_seq * nh);
  hipLaunchKernelGGL(k_attn, g, dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_embed: unfold(2,65,1) -> Conv2d(1,4,(1,5)) -> Linear(244,256) -> *16 + pos_embedding_freq
//                                                                             amt_apc.py:79-109
// There is no non-linearity between the conv and the linear, so they are folded on the host into
// one [256][65] map (K padded to 80).  Workgroup: 32 bins x 64 frames; each wave 16 frames.
// ================================================================================================
#define LDE 88    // folded-weight LDS row stride (elements): 176 B
#define EFB 64    // frames per workgroup
#define ELDX 33   // spec tile row stride (floats): [time][32 bins + 1]
#define ELDP 260  // pos tile row stride (elements)

__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[256 * LDE * 2 + (EFB + 80) * ELDX * 4 + 32 * ELDP * 2 + 256 * 4];
  e16* Wsm = reinterpret_cast<e16*>(smem);
  float* Xsm = reinterpret_cast<float*>(smem + 256 * LDE * 2);
  e16* Psm = reinterpret_cast<e16*>(smem + 256 * LDE * 2 + (EFB + 80) * ELDX * 4);
  float* bsm = reinterpret_cast<float*>(smem + 256 * LDE * 2 + (EFB + 80) * ELDX * 4 + 32 * ELDP * 2);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * 32;
  const int fl0 = blockIdx.y * EFB;              // first frame of this block inside the chunk
  const int wl = blockIdx.z, w = a.w0 + wl;

  // folded weights [256][80] -> LDS (10 chunks of 16 B per row)
  for (int c = tid; c < 256 * 10; c += 256) {
    const int row = c / 10, ch = c - row * 10;
    *reinterpret_cast<uint4*>(Wsm + row * LDE + ch * 8) = *reinterpret_cast<const uint4*>(a.Wf + row * 80 + ch * 8);
  }
  bsm[tid] = a.bf[tid];
  // pos rows for the 32 bins
  for (int c = tid; c < 32 * 32; c += 256) {
    const int row = c >> 5, ch = c & 31;
    *reinterpret_cast<uint4*>(Psm + row * ELDP + ch * 8) = *reinterpret_cast<const uint4*>(a.pos + (long long)(b0 + row) * 256 + ch * 8);
  }
  // spec tile: times (f0+fl0) .. +EFB+80 of window w, 32 bins; centred
  const int t_base = a.f0 + fl0;
  for (int c = tid; c < (EFB + 80) * 32; c += 256) {
    const int tr = c >> 5, bin = c & 31;
    const int t = t_base + tr;                     // time index inside the window's input [0, nf + 2*margin)
    float v = 0.f;
    if (t < a.nf + 2 * a.margin) {
      if (a.feat_mode) {
        const long long g = (long long)w * a.nf + t - a.margin;
        v = (g >= 0 && g < a.T) ? a.src[g * a.s_t + (long long)(b0 + bin) * a.s_bin] : a.pad_value;
      } else {
        v = a.src[(long long)w * a.s_win + (long long)(b0 + bin) * a.s_bin + (long long)t * a.s_t];
      }
      v -= a.center;
    }
    Xsm[tr * ELDX + bin] = v;
  }
  __syncthreads();

  for (int fi = 0; fi < EFB / 4; ++fi) {
    const int fl = wave * (EFB / 4) + fi;          // frame inside the block
    if (fl0 + fl >= a.fc) break;                   // wave-uniform
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      e16x8 xf;
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = (e16)Xsm[(fl + s * 16 + h * 8 + j) * ELDX + r];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const e16x8 wf = *reinterpret_cast<const e16x8*>(Wsm + (t * 32 + r) * LDE + s * 16 + h * 8);
        acc[t] = mfma32(wf, xf, acc[t]);
      }
    }
    e16* yrow = a.Y + ((long long)(wl * a.fc + fl0 + fl) * 256 + b0 + r) * 256;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = t * 32 + 8 * q + 4 * h;
        const e16x4 p = *reinterpret_cast<const e16x4*>(Psm + r * ELDP + f);
        *reinterpret_cast<e16x4*>(yrow + f) =
            pack4e((acc[t][4 * q + 0] + bsm[f + 0]) * 16.f + bf2f(p[0]), (acc[t][4 * q + 1] + bsm[f + 1]) * 16.f + bf2f(p[1]),
                  (acc[t][4 * q + 2] + bsm[f + 2]) * 16.f + bf2f(p[2]), (acc[t][4 * q + 3] + bsm[f + 3]) * 16.f + bf2f(p[3]));
      }
  }
}

int launch_embed(const EmbedArgs& a, hipStream_t st) {
  if (a.fc <= 0 || a.n_win <= 0 || a.margin != 32) ETD_FAIL(ETD_EINVAL, "embed: bad args");
  ETD_LAUNCH_FILTER("k_embed");
  ProfScope ps("k_embed", st, 2.0 * a.n_win * a.fc * 256.0 * 256 * 65, (double)a.n_win * a.fc * 256 * 256 * 2);
  dim3 g(8, (a.fc + EFB - 1) / EFB, a.n_win);
  hipLaunchKernelGGL(k_embed, g, dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// k_heads: onset/offset/mpe = sigmoid(Linear(256,1)) in fp32, velocity = argmax(Linear(256,128))
//                                           amt_apc.py:186-189,217-220 + extractor.py:242,248
// Swapped orientation: a token's 128 velocity logits sit in two lanes (l, l^32) -> argmax is
// lane-local + one exchange.  Ties resolve to the lowest index like torch.argmax.
// ================================================================================================
__global__ __launch_bounds__(256) void k_heads(HeadsArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(128 + 160) * LDK * 2 + 160 * 4];
  e16* Xs = reinterpret_cast<e16*>(smem);
  e16* Ws = Xs + 128 * LDK;
  float* sb = reinterpret_cast<float*>(smem + (128 + 160) * LDK * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128;
  if (tid < 160) sb[tid] = a.bias[tid];
  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
      *reinterpret_cast<uint4*>(Xs + row * LDK + ch * 8) = *reinterpret_cast<const uint4*>(a.X + (long long)gm * 256 + kc * 64 + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(Ws + row * LDK + ch * 8) = *reinterpret_cast<const uint4*>(a.W + (long long)row * 256 + kc * 64 + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const e16x8 xf = *reinterpret_cast<const e16x8*>(Xs + (wave * 32 + r) * LDK + s * 16 + h * 8);
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const e16x8 wf = *reinterpret_cast<const e16x8*>(Ws + (t * 32 + r) * LDK + s * 16 + h * 8);
        acc[t] = mfma32(wf, xf, acc[t]);
      }
    }
    __syncthreads();
  }
  const int m = m0 + wave * 32 + r;
  // velocity argmax over this lane's 64 logits, then against the other half
  float best = -INFINITY; int bidx = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = t * 32 + acc_row(i, h);
      const float v = acc[t][i] + sb[n];
      acc[t][i] = v;
      if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
    }
  const float ob = xhalf(best);
  const int oi = __shfl_xor(bidx, 32, 64);
  if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  if (m < a.M) {
    long long oidx;
    if (a.time_layout) {
      const int per_w = a.nn * a.nf;
      const int w = m / per_w, rem = m - w * per_w, note = rem / a.nf, f = rem - note * a.nf;
      oidx = ((long long)w * a.nf + f) * a.nn + note;
    } else {
      oidx = m;
    }
    oidx += a.out_off;
    if (h == 0) {
      a.vel[oidx] = (int8_t)bidx;
      const float lo = acc[4][0] + sb[128], lf = acc[4][1] + sb[129], lm = acc[4][2] + sb[130];
      a.onset[oidx] = 1.f / (1.f + expf(-lo));
      a.offset[oidx] = 1.f / (1.f + expf(-lf));
      a.mpe[oidx] = 1.f / (1.f + expf(-lm));
    }
    if (a.vel_logit) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a.vel_logit[oidx * 128 + t * 32 + acc_row(i, h)] = acc[t][i];
    }
  }
}

int launch_heads(const HeadsArgs& a, hipStream_t st) {
  if (a.M <= 0) ETD_FAIL(ETD_EINVAL, "heads: bad M");
  ETD_LAUNCH_FILTER("k_heads");
  ProfScope ps("k_heads", st, 2.0 * a.M * 256 * 131, (double)a.M * 256 * 2);
  hipLaunchKernelGGL(k_heads, dim3((a.M + 127) / 128), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
// freq-major decoder state -> time-major time-decoder input, *sqrt(256) + pos_embedding_time
//                                                                             amt_apc.py:203-205
__global__ void k_freq2time(const e16* __restrict__ src, e16* __restrict__ dst, const float* __restrict__ pos,
                            int nw, int fc, int f0, int nf, int nn) {
  const long long total = (long long)nw * fc * nn * 32;           // 16-byte chunks
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int ch = (int)(i & 31);
    const long long row = i >> 5;                                 // (wl*fc + fl)*nn + note
    const int note = (int)(row % nn);
    const long long fr = row / nn;
    const int fl = (int)(fr % fc), wl = (int)(fr / fc);
    const e16x8 v = *reinterpret_cast<const e16x8*>(src + row * 256 + ch * 8);
    const float* pp = pos + (long long)(f0 + fl) * 256 + ch * 8;
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (e16)(bf2f(v[j]) * 16.f + pp[j]);
    *reinterpret_cast<e16x8*>(dst + (((long long)wl * nn + note) * nf + f0 + fl) * 256 + ch * 8) = o;
  }
}
int launch_freq2time(const e16* src, e16* dst, const float* pos, int nw, int fc, int f0, int nf, int nn, hipStream_t st) {
  const long long total = (long long)nw * fc * nn * 32;
  ETD_LAUNCH_FILTER("k_freq2time");
  ProfScope ps("k_freq2time", st, 0, (double)total * 32);
  long long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_freq2time, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, pos, nw, fc, f0, nf, nn);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

// ================================================================================================
__global__ void k_f32_to_bf16(const float* __restrict__ s, e16* __restrict__ d, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = (e16)s[i];
}
int launch_f32_to_bf16(const float* src, e16* dst, long long n, hipStream_t st) {
  if (n <= 0) return ETD_OK;
  long long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
