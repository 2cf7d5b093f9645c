// Kernel argument structs + launchers for the hFT-Transformer (Extract stage) kernels.
#pragma once
#include "common.h"
#include "dec_kernels.h"

// ---- generic token-major linear layer:  Y = epi(X[M,K] * W[N,K]^T + bias) -----------------------
// All activations are bf16 row-major with 256-multiple feature counts; weights keep nn.Linear's
// [out][in] layout (K contiguous), which is exactly the MFMA operand layout for both operands.
// Weight layout for k_linear's decoder modes (launch_linear_dec): MFMA-fragment order.  Block (t, s) -- n-tile t of 32 output features, k-step s of 16 inputs --
// is 512 elements at ((t * (K / 16) + s) * 512); inside it lane l (0..63) owns 8 consecutive elements: row 32 t + (l & 31),
// columns 16 s + 8 (l >> 5) .. + 8.  Same element count as [N][K]; feature-block and z-batch offsets (n0 * K, z * N * K)
// are unchanged.  N % 32 == 0, K % 16 == 0.
// The extractor's 16-bit operand type.  IEEE half by default: on gfx950 v_mfma_f32_32x32x16_f16 has the layout and the rate of the bf16 form, and its 11
// significant bits (against 8) cut the extractor's distance to the fp32 reference by ~6x at no cost (measured: profiles/r05_ext_f16.txt; why the range suffices:
// DESIGN section 2).  -DETD_EXT_BF16 builds the bf16 extractor of rounds 1-4 from the same sources.  The EtudeDecoder's 16-bit serving mode is IEEE half as well (d16, dec_kernels.h).
#ifdef ETD_EXT_BF16
typedef bf16 e16; typedef bf16x8 e16x8; typedef bf16x4 e16x4; typedef bf16x2 e16x2;
#ifdef __HIPCC__
__device__ __forceinline__ e16x4 pack4e(float a, float b, float c, float d) { return pack4(a, b, c, d); }
#endif
#define ETD_EXT_IS_F16 0
#else
typedef f16 e16; typedef f16x8 e16x8; typedef f16x4 e16x4; typedef f16x2 e16x2;
#ifdef __HIPCC__
__device__ __forceinline__ e16x4 pack4e(float a, float b, float c, float d) { return pack4h(a, b, c, d); }
#endif
#define ETD_EXT_IS_F16 1
#endif
#include <cstdint>
#include <vector>
inline void pack_wfrag_host(const uint16_t* src, int N, int K, uint16_t* dst) {
  const int kb = K / 16;
  for (int t = 0; t < N / 32; ++t)
    for (int sx = 0; sx < kb; ++sx)
      for (int l = 0; l < 64; ++l) {
        const uint16_t* sp = src + (size_t)(t * 32 + (l & 31)) * K + sx * 16 + (l >> 5) * 8;
        uint16_t* dp = dst + (((size_t)t * kb + sx) * 64 + l) * 8;
        for (int e = 0; e < 8; ++e) dp[e] = sp[e];
      }
}

struct LinArgs {
  const e16* X; int ldx;            // [M, K]
  const e16* W;                     // [N, K] weights: row-major for the extractor modes (0, 1, 2); FRAGMENT ORDER (pack_wfrag_host above) for the decoder modes (launch_linear_dec)
  const float* bias;                 // [N]
  int M, N, K;                       // K % 64 == 0, N % 256 == 0
  e16* Y; int ldy;                  // row-major destination for n-blocks < vt_block (may be null if all go to VT)
  int nb0, nby;                      // (set by the launcher) first 256-feature block of this launch, number of blocks
  int vt_block;                      // blockIdx.y == vt_block -> that 256-feature block is stored TRANSPOSED (V^T); -1 none
  e16* VT; int S, Spad;             // VT[((seq*4+head)*64+d)*Spad + pos], seq = m / S, pos = m % S
  int relu;
  long long* tbuf;                   // ETD_LIN_STAMP: per-workgroup clock64 stamps [grid][16] (wave 0)
  int dbg;                           // timing ablations (ETD_LIN_DBG): 1 = skip the epilogue, 2 = skip the K loop (results are garbage); 4 = V^T block with direct 8-byte stores (correct, slower)
  // z-batching (several weight sets over the same X): per-blockIdx.z element offsets
  long long wz, bz, yz, vtz;
  // LayerNorm epilogue (N == 256): Y = LN(acc + bias + R) * gamma + beta
  const e16* R; int ldr; int r_mod;  // residual row = r_mod > 0 ? m % r_mod : m
  const float* gamma; const float* beta;
  // decoder epilogues on the same 128x256 tile (X = e16 [M,K], W = e16 [N,K]): dec_epi = DEPI_* (dec_kernels.h)
  DGemmArgs dec;
};
int launch_linear_dec(const LinArgs& a, int dec_epi, hipStream_t st);
int launch_linear(const LinArgs& a, int nz, hipStream_t st);
int launch_linear_ln(const LinArgs& a, hipStream_t st);

// ---- multi-head attention, head_dim 64, 4 heads, non-causal -------------------------------------
struct AttnArgs {
  const e16* Q; int ldq; long long q_seq_stride;   // Q row = seq*q_seq_stride/ldq.. (elements): Q + seq*q_seq_stride + q*ldq + head*64
  const e16* K; int ldk; long long k_seq_stride;   // K + seq*k_seq_stride + key*ldk + head*64
  const e16* VT; int Spad;                          // VT + ((seq*4+head)*64 + d)*Spad + key
  e16* O; int ldo; long long o_seq_stride;          // O + seq*o_seq_stride + q*ldo + head*64
  int n_seq, Sq, Sk;
  float scale_log2e;                                 // (1/sqrt(64)) * log2(e)
  int n_heads;                                       // 0 -> 4 (the hFT model)
  // ragged causal batches (EtudeDecoder prefill): sequence s covers rows [seq_row0[s], +seq_len[s]) of Q/K/O; key j visible to query i iff j <= i
  const int* seq_row0; const int* seq_len; int causal;
  double flops_hint;                                 // algorithmic FLOPs of this launch (profiler only; 0 = derive from Sq/Sk)
};
int launch_attn(const AttnArgs& a, hipStream_t st);
struct EmbedArgs;

// ---- front-end embedding: unfold(65) + conv(1x5) + Linear(244,256) folded to one 65->256 map ----
struct EmbedArgs {
  const float* src;            // features
  long long s_win, s_bin, s_t; // element strides of src for (window, bin, time-in-window)
  int feat_mode;               // 1: src is [T][n_bin] features, time t of window w is frame w*nf + t - margin (pad value outside [0,T))
  long long T;                 // valid frames (feat_mode)
  float pad_value;             // -18.0
  float center;                // value subtracted before bf16 rounding (folded into bias)
  const e16* Wf;              // [256][80] folded weights (K padded 65->80 with zeros)
  const float* bf;             // [256] folded bias (incl. center * sum_t Wf)
  const e16* pos;             // [256 bins][256] e16 pos_embedding_freq
  e16* Y;                     // [(w_local*fc + f_local)*256 + bin][256]
  int w0, n_win;               // first window, number of windows in this launch
  int f0, fc;                  // first frame within the window, frames in this chunk
  int nf, margin;
};
int launch_embed(const EmbedArgs& a, hipStream_t st);

// ---- output heads: 3 x Linear(256,1)+sigmoid (fp32) and Linear(256,128)+argmax -> int8 ----------
struct HeadsArgs {
  const e16* X;               // [M, 256]
  const e16* W;               // [160][256]: rows 0..127 velocity, 128 onset, 129 offset, 130 mpe, rest 0
  const float* bias;           // [160]
  int M;
  int time_layout;             // 1: m=(w*nn+note)*nf+f -> out[(w*nf+f)*nn+note];  0: out[m]
  int nf, nn;
  long long out_off;           // element offset added to the output index
  float* onset; float* offset; float* mpe; int8_t* vel;
  float* vel_logit;            // optional debug: [.. ][128] fp32 logits (null in production)
};
int launch_heads(const HeadsArgs& a, hipStream_t st);

// freq-major [ (wl*fc+fl)*nn + note ][256] -> time-major TI[ (wl*nn+note)*nf + f0+fl ][256] = x*16 + pos_time[f0+fl]
int launch_freq2time(const e16* src, e16* dst, const float* pos, int nw, int fc, int f0, int nf, int nn, hipStream_t st);

// small utilities
int launch_f32_to_bf16(const float* src, e16* dst, long long n, hipStream_t st);

// ---- fused position-wise feed-forward sub-layer (csrc/ext_fused.hip): Y = LN(X + relu(X W1^T + b1) W2^T + b2) * gamma + beta
struct FfnArgs {
  const e16* X;               // [M][256] bf16 row-major (also the residual)
  const e16* Wf;              // packed weight stream (pack_ffn_weights): [16][32][64][8]
  const float* b1;             // [512]
  const float* b2; const float* gamma; const float* beta;   // [256] each
  e16* Y;                     // [M][256]; may alias X (a token's row is read completely before it is written, by the same lane pair)
  int M;
};
int launch_ffn_fused(const FfnArgs& a, hipStream_t st);
void pack_ffn_weights(const float* W1, const float* W2, uint16_t* dst, uint16_t (*f2bf)(float));

// ---- K = 256 projections on the fused skeleton (csrc/ext_fused.hip: k_proj256)
enum { PROJ_ROW = 0, PROJ_VT = 1, PROJ_LN = 2, PROJ_KFRAG = 3, PROJ_VFRAG = 4 };
#define PROJ_MAX_BLOCKS 6
struct ProjBlock {
  const e16* Wf;              // packed [256 out][256 in] block (pack_proj_weights; rows permuted for ROW / LN, natural for VT)
  const float* bias;           // [256]
  int kind;                    // PROJ_ROW / PROJ_VT / PROJ_LN
  int relu;                    // ROW only
  e16* dst; int ldd;          // ROW / LN: dst[m * ldd + feature] (col offset folded into dst); VT: V^T base (z offset folded in)
};
struct ProjArgs {
  const e16* X; int ldx; int M;      // [M][256] bf16 rows, row stride ldx
  int nblk; ProjBlock blk[PROJ_MAX_BLOCKS];
  int S, Spad;                        // VT blocks: seq = m / S, pos = m % S; row stride of V^T
  int kv_nstep;                       // KFRAG / VFRAG blocks: S / 64; dst = fragment images [(seq * 4 + head)][kv_nstep][8192] (k_attn_frag)
  const e16* R; int r_mod;           // LN blocks: residual rows [.][256] (row = r_mod > 0 ? m % r_mod : m)
  const float* gamma; const float* beta;
};
int launch_proj256(const ProjArgs& a, hipStream_t st);
void pack_proj_weights(const float* W, bool permute_rows, uint16_t* dst, uint16_t (*f2bf)(float));

// ---- one whole hFT EncoderLayer (amt_apc.py:236-259) per 256-token sequence in ONE launch (csrc/ext_fused.hip: k_enc_layer):
// QKV projection, 4-head attention, output projection + LayerNorm, feed-forward + LayerNorm; X is read once, Y written once
struct EncLayerArgs {
  const e16* X;               // [n_seq * 256][256]
  const e16* Wl;              // packed layer stream (pack_enc_layer_weights): 32 chunks of 32 KiB
  const float* bqkv;           // [768] q | k | v biases
  const float* bo; const float* gamma; const float* beta;   // [256]
  const float* b1; const float* b2;                         // [512], [256]
  e16* Y;                     // [n_seq * 256][256]; may alias X
  int n_seq;
};
int launch_enc_layer(const EncLayerArgs& a, hipStream_t st);
// Wq, Wk, Wv, Wo: [256][256]; W1 [512][256]; W2 [256][512] (fp32, nn.Linear layout) -> dst[32 * 16384] e16
void pack_enc_layer_weights(const float* Wq, const float* Wk, const float* Wv, const float* Wo, const float* W1, const float* W2, uint16_t* dst, uint16_t (*f2bf)(float));

// ---- the part of a layer behind its attention in one launch (csrc/ext_fused.hip: k_post_attn):
//      x1 = LN(R + AO Wo^T + bo) and, when Wffn is set, y = LN(x1 + FFN(x1)) with the same (shared) LayerNorm
struct PostAttnArgs {
  const e16* AO;              // [M][256] attention output
  const e16* R; int r_mod;    // residual rows (row = r_mod > 0 ? m % r_mod : m)
  const e16* Wo;              // fc_o as a k_proj256 block (pack_proj_weights, rows permuted)
  const e16* Wffn;            // k_ffn_fused's stream, or null: stop after the first LayerNorm
  const float* bo; const float* gamma; const float* beta; const float* b1; const float* b2;
  e16* Y; int M;
};
int launch_post_attn(const PostAttnArgs& a, hipStream_t st);      // (-DETD_EXPERIMENTS builds only)

// ---- attention over K / V fragment images (csrc/ext_fused.hip: k_attn_frag); 4 heads x 64
struct AttnFragArgs {
  const e16* Q; int ldq; long long q_seq_stride;   // Q + seq * q_seq_stride + q * ldq + head * 64
  const e16* KV;                                    // [(seq * 4 + head)][Sk / 64][8192]: k_proj256's KFRAG | VFRAG images
  e16* O; int ldo; long long o_seq_stride;          // O + seq * o_seq_stride + q * ldo + head * 64
  int n_seq, Sq, Sk;                                 // Sk % 64 == 0
  float scale_log2e;
};
int launch_attn_frag(const AttnFragArgs& a, hipStream_t st);
