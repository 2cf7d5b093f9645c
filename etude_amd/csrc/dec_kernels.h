// Kernel argument structs + launchers for the EtudeDecoder (GPT-NeoX) decode path.
#pragma once
#include "common.h"

// The EtudeDecoder's 16-bit operand type: weights, LayerNorm rows, GELU / attention outputs and the KV cache of the serving mode (precision 1).  IEEE half by default:
// on gfx950 v_mfma_f32_32x32x16_f16 has the layout and the rate of the bf16 form, the bytes in HBM are the same, and its 11 significant bits (against 8) put the
// serving mode's logits 8x closer to the fp32 reference (measured: profiles/r05_dec_f16.txt; why the range suffices: DESIGN section 2).  -DETD_DEC_BF16 builds the
// bf16 decoder of rounds 1-4 from the same sources.
#ifdef ETD_DEC_BF16
typedef bf16 d16; typedef bf16x8 d16x8; typedef bf16x4 d16x4; typedef bf16x2 d16x2;
#define ETD_MFMA16_DEC "v_mfma_f32_32x32x16_bf16"
#define ETD_DEC_IS_F16 0
#ifdef __HIPCC__
__device__ __forceinline__ d16x4 pack4d(float a, float b, float c, float d) { return pack4(a, b, c, d); }
#endif
#else
typedef f16 d16; typedef f16x8 d16x8; typedef f16x4 d16x4; typedef f16x2 d16x2;
#define ETD_MFMA16_DEC "v_mfma_f32_32x32x16_f16"
#define ETD_DEC_IS_F16 1
#ifdef __HIPCC__
__device__ __forceinline__ d16x4 pack4d(float a, float b, float c, float d) { return pack4h(a, b, c, d); }
#endif
#endif

// Row counts: a batched PREFILL of more than DS_MAX_ROWS rows runs on the big-tile MFMA GEMMs, anything smaller on the weight-streaming
// skinny tile; a decode STEP stays on the fused step kernels (skinny tiles, split-K slabs) up to DS_STEP_MAX_ROWS rows -- 1 728 streams is
// BASELINE configs[4]'s whole 64-clip x 27-tuple grid in one launch per kernel.
#define DS_MAX_ROWS 512
#define DS_STEP_MAX_ROWS 2048

// per-row metadata of a forward pass over M rows (prefill: one stream, T rows; decode: one row per stream)
struct DecRows {
  const int* slot;     // [M] KV slot of the row
  const int* pos;      // [M] position (= index of this token in the slot's KV cache)
  const int* active;   // [M] 0 -> the row must not write KV / state (finished stream)
  const int* seq;      // [M] index of the row's prompt inside a batched prefill (only read when the prefill scratch is set)
};

// temperature / top-p sampling (etude_decoder.py:321-331).  Lives in DEVICE memory so that captured graphs follow a change;
// inv_temp <= 0 selects greedy argmax.  A draw is a pure function of (seed, the stream's key, tokens generated so far in the
// bar): results do not depend on which slot or engine a job lands on.
struct DSampleCfg { float inv_temp; float top_p; unsigned long long seed; };

// device-side span stamps of the attention launches (etd_decoder_stamp): u64 words [0] sum of spans, [1] launches folded in, [2 + bank] start, [8 + 64 bank + i] end slots,
// then ETD_STAMP_LOGCAP (start, end) pairs of the folded launches in order -- the union of several engines' launches is formed from those on the host
#define ETD_STAMP_HDR (8 + 2 * 64)
#define ETD_STAMP_LOGCAP 131072

enum { DEPI_BIAS = 0, DEPI_GELU = 1, DEPI_RESID = 2, DEPI_LOGITS = 3, DEPI_QKV = 4, DEPI_PARTIAL = 5, DEPI_RELU = 6 /* k_gemm3 only */ };

struct DGemmArgs {
  const float* X; int ldx;       // [M, K] fp32 activations
  const void* W;                 // [Npad, K] weights (float or d16), K contiguous
  const void* Wf;                // optional: the same d16 weights in MFMA-fragment order (pack_wfrag_host) -- k_dstep_qkv_up_mt
  const float* bias;             // [Npad] or null
  int M, N, K;                   // N = valid output features (stores guarded), Npad % 128 == 0
  int Npad;
  // LayerNorm prologue over K (only K == hidden): xhat = (x-mean)*rstd*g + b
  const float* ln_g; const float* ln_b; float ln_eps;
  float* Y; int ldy;             // BIAS / GELU / LOGITS destination
  d16* Yb;                      // if set, BIAS / GELU store d16 here instead (row stride ldy)
  int k_splits;                  // DEPI_PARTIAL (skinny kernel only): K is split over this many workgroups (grid.z); slice z stores its
                                 // raw fp32 partial products to Y + z*M*ldy, k_resid_ln_rows adds them up in a fixed order
  const d16* Xb;                // if set, the input is d16 [M, K] (already LayerNorm'ed), row stride ldx
  // RESID: hout = (acc + bias + add[m][n]) + h[m][n]
  const float* add; const float* hin; float* hout;
  // QKV: rope + scatter
  DecRows rows;
  const float* rope_cos; const float* rope_sin;   // [max_ctx][rot/2]
  int rot_half;                  // rotary_ndims / 2 (8)
  float* Q;                      // [M][hidden]
  void* Kc; void* Vc;            // KV cache base of this LAYER: [slot][head][max_ctx][64]
  long long slot_stride;         // elements between slots
  int max_ctx, n_heads;
  // batched prefill on the MFMA flash-attention kernel (null outside the d16 big-M path):
  d16* Qb;                      // [M][hidden] d16 row-major RoPE'd Q (K / V are read from the cache rows: k_pattn)
  // exact-parity mode on the f16 matrix cores (csrc/gemm3.h): the fp32 weights again as hi / lo f16 planes in k_gemm3's streaming order
  const void* Wp; int w_log2;    // planes and log2 of the power-of-two scale they carry
  int x_log2;                    // log2 of the scale X is split at (|X| 2^x_log2 < 2^15 by a provable bound)
};
int launch_dgemm(const DGemmArgs& a, int epi, bool w_bf16, hipStream_t st);
// decode step, d16: fused-QKV(+RoPE, KV append) and MLP-up(+GELU) projections of one layer in a single launch
int launch_dstep_qkv_up(const DGemmArgs& q, const DGemmArgs& up, hipStream_t st);

// decode step, d16: final LayerNorm + lm_head + greedy argmax + stream state update, then the NEXT step's token embedding,
// row metadata and first-layer LayerNorms -- the tail of step t and the head of step t+1 in one launch
struct DHeadArgs {
  const float* hfin; int M, H, V, Vpad;       // last layer's residual stream [M][H]
  const float* lnf_g; const float* lnf_b; float eps;
  const d16* Whead;                           // lm_head in fragment order [Vpad/32][H/16][64][8] (api_dec.hip: head_frag)
  int* row_slot; int* row_pos; int* row_active; int* row_sp;   // in: this step's rows; out: next step's (row_sp: (slot, pos) pairs)
  int* cur_tok; int* len; int* done; int* n_out; int* out_tok; int out_cap; const int* eos; const int* limit;
  const int* tgt_attrs; const float* tgt_proj /* [slot][H]: attribute projection of the target attributes */; int tgt_cls; int n_bins;
  const float* word; const float* cls_emb; const float* attr_tab;
  const float* g1; const float* b1; const float* g2; const float* b2;   // layer 0 LayerNorms
  float* h; d16* x1; d16* x2;                // next step's embeddings [M][H] and their LayerNorms
  const DSampleCfg* samp; const unsigned long long* rng_key;   // [slots]; samp == null -> greedy
  float* logits_dbg;                           // test hook (etd_debug_decoder_step_logits): [M][V] logits of this step, null in production graphs
};
int launch_dstep_head(const DHeadArgs& a, hipStream_t st);

struct DAttnArgs {
  const float* Q;                // [M][hidden]
  const void* Kc; const void* Vc; long long slot_stride; int max_ctx, n_heads;
  DecRows rows; int M;
  const int* row_sp;             // optional [M][2] = (slot, position) pairs: one scalar load instead of two dependent ones
  int identity;                  // row i uses slot i (host knowledge): the first K/V block is requested before the row metadata arrives
  int pair;                      // k_dstep_attn_down: two rows of a head per 8-wave workgroup (host decision by mean context, api_dec.hip)
  float* O;                      // [M][hidden]
  d16* Ob; int ldob;            // optional d16 copy of O (input of the dense GEMM in the d16 pipeline), row stride ldob (0 = hidden)
  float scale;
  double bytes_hint;             // algorithmic K+V bytes this launch reads (host estimate, profiler only)
  // k_dstep_attn_down only: attention.dense applied inside the attention workgroup
  const d16* dense_w;           // [heads][512][64]: per head, the [out][64] slice of attention.dense (contiguous 64 KiB)
  float* dense_out;              // [heads][M][512] fp32 partial sums = split-K slabs of k_resid_ln_rows
  float* dbg;                    // diagnostic (step trace): [heads][M][256 lanes][8] = lr, mr, o[0] after the key loop, lr after merge stages 8 / 16 / 32
  unsigned long long* stamp;     // k_dstep_attn_down, measurement (null in production graphs): device-side span accumulator, layout at the kernel
  int stamp_par;                 // bank of this launch (layer & 1)
};
int launch_dattn(const DAttnArgs& a, bool kv_bf16, hipStream_t st);
// k_dstep_attn_down with the row kernel folded in: every contributor of a row's split-K slabs (its 8 attention workgroups, the
// 16 x k_splits down-projection units of its row tile) stores its slab write-through and adds 1 to cnt[row]; the one that
// brings the count to `target` is the last, reads the row's slabs back and does what k_resid_ln_rows does for that row
// (same additions in the same order: bit-identical).  cnt is all zero between launches (the last arriver resets its word).
struct DRowFin {
  int* cnt; int target;          // [M] arrival counters of this layer; target = n_heads + 16 * k_splits
  const float* P; int nslab;     // slabs [nslab][M][512]: k_splits of the down projection, then one per head
  const float* bias; const float* hin; float* hout;
  const float* g1; const float* b1; const float* g2; const float* b2; float eps;   // next layer's LayerNorms (x1 == null: none)
  d16* x1; d16* x2;
};
// decode step, d16: attention of every (row, head) WITH its share of attention.dense, and -- in the same launch, on other
// workgroups -- the MLP down projection (which depends on the QKV|up launch only): `down` is a DEPI_PARTIAL request
// (k_splits slabs of K / k_splits each, over the first K columns of the (down | dense) weight).
int launch_dstep_attn_down(const DAttnArgs& a, const DGemmArgs& down, const DRowFin* fin, hipStream_t st);

// batched prefill, d16 (csrc/dec_prefill.hip): ragged causal MFMA flash attention over all prompts of a pass, K / V read from the KV cache rows the QKV
// epilogue has just written (no scratch copies)
struct PAttnArgs {
  const d16* Q; int ldq;              // [M][hidden] RoPE'd queries, d16 row-major (row stride ldq)
  const d16* Kc; const d16* Vc;      // this LAYER's cache: [slot][head][max_ctx][64]
  long long slot_stride; int max_ctx, n_heads;
  d16* O; int ldo;                    // [M][..] attention output rows (row stride ldo)
  const int* seq_row0; const int* seq_len; const int* row_slot;   // prompt s covers rows [seq_row0[s], + seq_len[s]); its slot = row_slot[seq_row0[s]]
  int n_seq, max_len;
  float scale_log2e;                   // (1 / sqrt(64)) * log2(e)
  double flops_hint;
};
int launch_pattn(const PAttnArgs& a, hipStream_t st);

// batched prefill, d16 (csrc/dec_prefill.hip): the fused QKV projection with the token block stationary in registers (hidden 512, 8 heads): RoPE'd Q rows
// to Qb, K / V rows to the cache
struct PQkvArgs {
  const d16* X; int ldx;              // [M][512] input_layernorm(h) (d16)
  const d16* Wf;                      // query_key_value.weight [1536][512] in MFMA-fragment order (pack_wfrag_host)
  const float* bias;                   // [1536]
  int M, N;
  DecRows rows;
  const float* rope_cos; const float* rope_sin;   // [max_ctx][8]
  d16* Qb;                            // [M][512]
  d16* Kc; d16* Vc; long long slot_stride; int max_ctx, n_heads;
};
int launch_pqkv(const PQkvArgs& a, hipStream_t st);

// h fp32 [M,H] -> LayerNorm with (g1,b1) and (g2,b2) -> two d16 matrices (the two parallel-residual branches read the same h)
int launch_ln_rows(const float* h, int M, int H, const float* g1, const float* b1, const float* g2, const float* b2, float eps,
                   d16* x1, d16* x2, hipStream_t st);
// hout = ((sum_z P[z] + bias) + add) + hin   [M][H] fp32, then (optionally) LayerNorm of hout with (g1,b1) / (g2,b2) -> d16 x1 / x2
int launch_resid_ln_rows(const float* P, int k_splits, const float* bias, const float* add, const float* hin, float* hout, int M, int H,
                         const float* g1, const float* b1, const float* g2, const float* b2, float eps, d16* x1, d16* x2, hipStream_t st);
// out[i][:] = src[idx[i]][:]  (fp32 rows of H)
int launch_gather_rows(const float* src, const int* idx, int n, int H, float* out, hipStream_t st);

// batched prefill, d16 (csrc/dec_fused.hip): MLP branch + attention.dense + parallel residual + the next layer's LayerNorms in
// one launch, bit-identical to k_linear<GELU> -> k_linear<RESID> -> k_ln_rows.  H = 512, I = 2048 only.
#define DMLP_NCHUNK 73                               // 64 KiB chunks of the weight stream: 65 of the MLP, 8 of attention.dense
#define DMLP_STREAM_ELEMS (DMLP_NCHUNK * 32 * 1024)  // d16 elements per layer
struct DMlpArgs {
  const d16* X2;                // [M][512] post_attention_layernorm(h_in)
  const d16* AO; int ldao;      // [M][512] attention output, row stride ldao (it sits behind the hidden block of Xcat)
  const float* hin; float* hout; // [M][512] fp32 residual stream in / out (different buffers)
  const d16* Wm;                // pack_dmlp_weights
  const float* b_up;             // [2048]
  const float* b_cat;            // [512] dense_4h_to_h.bias + attention.dense.bias
  const float* g1; const float* b1; const float* g2; const float* b2; float eps;   // next layer's LayerNorms (nx1 == null: none)
  d16* nx1; d16* nx2;          // [M][512]
  int M;
};
int launch_dmlp_fused(const DMlpArgs& a, hipStream_t st);
void pack_dmlp_weights(const uint16_t* Wup_bf16 /* [2048][512] */, const uint16_t* Wcat_bf16 /* [512][2560] */, uint16_t* dst /* DMLP_STREAM_ELEMS */);

struct DEmbedArgs {
  const int* ids; const int* cls; const int* attrs;   // [M], [M], [4][M] (explicit mode)  -- or null:
  const int* cur_tok; const int* tgt_attrs;           // decode mode: ids = cur_tok[slot], cls = tgt_cls, attrs = tgt_attrs[slot][4]
  int tgt_cls;
  DecRows rows; int M, H, n_bins;
  const int* slots; const int* len; const int* done;  // decode mode, optional: derive the rows from the slot list and WRITE them to
  int* row_slot_out; int* row_pos_out; int* row_active_out; int* row_sp_out;   //   these arrays ((slot, pos) pairs too)
  const float* word; const float* cls_emb; const float* attr_tab;  // [V][H], [C][H], [4][n_bins][H] (+bias in tab 0)
  float* h;
};
int launch_dembed(const DEmbedArgs& a, hipStream_t st);

struct DArgmaxArgs {
  const float* logits; int ldl; int V; int M;
  DecRows rows;
  int* cur_tok; int* len; int* done; int* n_out; int* out_tok; int out_cap; const int* eos; const int* limit;
  int set_len_from_pos;          // prefill: len[slot] = pos + 1 of the row
  const DSampleCfg* samp; const unsigned long long* rng_key;   // [slots]; samp == null -> greedy
};
int launch_dargmax(const DArgmaxArgs& a, hipStream_t st);

