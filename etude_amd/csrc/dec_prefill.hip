// Batched prefill of the EtudeDecoder, bf16: ragged causal flash attention over every prompt of a begin_bars pass, reading K and V
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

#define PA_LDK 72     // K tile row stride (elements): 144 B, conflict-free ds_read_b128
#define PA_LDV 96     // V tile row stride (elements): 192 B -> key rows 4 h + q of a transposing read fall on bank groups 0, 192, 128, 64 (mod 256 B)

typedef __attribute__((ext_vector_type(4))) short pa_s16x4;
typedef __attribute__((address_space(3))) pa_s16x4* pa_lds_s16x4;

__global__ __launch_bounds__(256) void k_pattn(PAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * PA_LDK * 2 + 64 * PA_LDV * 2];
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + 64 * PA_LDK;
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

  const bf16* qp = a.Q + (r0 + qi) * a.ldq + head * 64;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + s * 16 + h * 8);

  const long long cbase = (long long)slot * a.slot_stride + (long long)head * a.max_ctx * 64;
  const bf16* kbase = a.Kc + cbase;
  const bf16* vbase = a.Vc + cbase;

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
  const bf16* vtr = Vs + (4 * h + tq) * PA_LDV + 16 * tg + 4 * tp;
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
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + r) * PA_LDK + s * 16 + h * 8);
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
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)sT[ks >> 1][8 * (ks & 1) + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const pa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4)(vtr + (ks * 16) * PA_LDV + dt * 32));
          const pa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pa_lds_s16x4)(vtr + (ks * 16 + 8) * PA_LDV + dt * 32));
          const __attribute__((ext_vector_type(8))) short v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = mfma32(__builtin_bit_cast(bf16x8, v8), pf, o[dt]);
        }
      }
    }
    __syncthreads();
  }

  lrun += xhalf(lrun);
  const float inv = 1.f / lrun;
  if (qvalid) {
    bf16* op = a.O + (r0 + q0 + r) * a.ldo + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + 8 * q + 4 * h;
        *reinterpret_cast<bf16x4*>(op + d) = pack4(o[dt][4 * q] * inv, o[dt][4 * q + 1] * inv, o[dt][4 * q + 2] * inv, o[dt][4 * q + 3] * inv);
      }
  }
}

int launch_pattn(const PAttnArgs& a, hipStream_t st) {
  if (a.n_seq <= 0 || a.max_len <= 0 || a.n_heads <= 0 || !a.Q || !a.Kc || !a.Vc || !a.O || !a.seq_row0 || !a.seq_len || !a.row_slot || a.max_len > a.max_ctx ||
      (a.ldq % 8) || (a.ldo % 4) || (((uintptr_t)a.Q | (uintptr_t)a.Kc | (uintptr_t)a.Vc) & 15) || ((uintptr_t)a.O & 7))
    ETD_FAIL(ETD_EINVAL, "pattn: bad arguments");
  ETD_LAUNCH_FILTER("k_attn_causal");
  ProfScope ps("k_attn_causal", st, a.flops_hint, ((double)a.n_seq * 4.0 * a.max_len * 64 * a.n_heads) * 2);
  dim3 g((a.max_len + 127) / 128, a.n_seq * a.n_heads);
  hipLaunchKernelGGL(k_pattn, g, dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
