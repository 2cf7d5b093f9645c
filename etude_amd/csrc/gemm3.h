// fp32-grade GEMM and attention on the f16 matrix cores: every fp32 operand x is carried as TWO f16 planes
//     hi = f16(s x),  lo = f16(s x - hi)          (s a power of two: exact; hi + lo holds 22-23 significant bits of s x)
// and an fp32 product tile is three v_mfma_f32_32x32x16_f16 into one fp32 accumulator:  lo*hi + hi*lo + hi*hi  (the lo*lo term is below
// fp32's own rounding).  f16 x f16 products are exact in fp32 and the accumulator is fp32, so what is lost against a k-ordered fp32 fmaf
// chain is the 2^-22 .. 2^-23 of the operand split -- measured (tools/ubench/mfma_f16_split.hip, profiles/r05_f16_split.txt): the same
// error against an fp64 product as the fp32 chain itself (1.1e-7 of sum|ab| at K = 512 both), f16 subnormal operands are NOT flushed;
// tools/diag_split_f16.py: the decoder oracle with every GEMM and both attention products replaced by this arithmetic has the logits error of
// the fp32 oracle (4.7e-6 vs 4.1e-6 against fp64 at T = 1024; 9.1e-5 vs 9.9e-5 on the context weights).  Rate: 3 MFMAs of the 2.5 PFLOP/s
// f16 pipe = 833 TFLOP/s of fp32-grade products against 157 TFLOP/s of v_mfma_f32_32x32x2_f32.
//
// f16 has 5 exponent bits: the planes hold s x with s chosen so that |s x| < 2^15 by a PROVABLE bound of |x| (weights: their own maximum at
// load time; activations: bounds from the LayerNorm parameters and weight norms, g3_bound_* below) -- nothing can overflow, and an element
// below 2^-3 of the plane's range still carries an absolute error of at most 2^-25 of it (subnormal lo).
//
// This is the arithmetic of the library's exact-parity (precision "fp32") mode for every dense contraction of >= G3_MIN_ROWS rows:
// Reference ops: F.linear in etude/models/amt_apc.py:322-392 and HF modeling_gpt_neox.py:195-281 (fp32, no autocast: etude_decoder.py:333).
#pragma once
#include "dec_kernels.h"

// ETD_NO_GEMM3 (measurement: the round-4 fp32-MFMA / VALU kernel sequence instead) is read ONCE per process, here, for every call site
bool g3_enabled();

#define G3_MIN_ROWS 513            // below: the weight-streaming fp32 kernels (k_dgemm_s / k_dgemv), where a 128-token tile would idle most of the chip

// weights [N][K] fp32 -> planes in the order k_gemm3 streams them: [Npad/128 tile][K/32 chunk][plane hi|lo][128 rows][32 k] f16.
// Returns log2 of the scale the planes carry (max |s w| in [2^14, 2^15)).
int g3_pack_weights_host(const float* W, int N, int Npad, int K, uint16_t* dst);
static inline size_t g3_packed_elems(int Npad, int K) { return (size_t)Npad * K * 2; }

// ---- bounds (host, load time).  LayerNorm output y = z g + b with sum z^2 <= n, |z_k| <= sqrt(n - 1):
float g3_bound_ln(const float* g, const float* b, int n);                                   // max_k sqrt(n-1) |g_k| + |b_k|
// rows of a linear fed by that LayerNorm: |W_j . y + c_j| <= sqrt(n) ||W_j o g||_2 + |W_j . b| + |c_j|  (Cauchy-Schwarz)
float g3_bound_linear_of_ln(const float* W, const float* c, int N, int K, const float* g, const float* b);
// rows of a linear fed by anything bounded elementwise by bx: ||W_j||_1 bx + |c_j|
float g3_bound_linear(const float* W, const float* c, int N, int K, float bx);
// the same bound per output row j (out[j]), for stacked / interleaved projections whose parts are consumed apart (GPT-NeoX's [head][q|k|v][64] rows)
void g3_row_bounds_of_ln(const float* W, const float* c, int N, int K, const float* g, const float* b, float* out);
// log2 of the power-of-two scale that keeps |s x| < 2^15 for |x| <= bound
int g3_scale_log2(float bound);

// fp32 LayerNorm rows (the GEMM's input is then plain fp32): x1 = LN(h; g1, b1), x2 = LN(h; g2, b2) (x2 / g2 may be null).  H % 256 == 0, H <= 1024.
int launch_ln_rows_f32(const float* h, int M, int H, const float* g1, const float* b1, const float* g2, const float* b2, float eps, float* x1, float* x2, hipStream_t st);

// Y = epi(X W^T): X fp32 [M][K] (row stride ldx), W as packed planes; the DGemmArgs epilogue fields as for launch_dgemm (fp32 outputs / fp32 KV rows).
// a.Wp = packed planes, a.w_log2 / a.x_log2 = the scales' logarithms.
int launch_gemm3(const DGemmArgs& a, int epi, hipStream_t st);
// the same for 2 .. 512 rows (weight-streaming shape: 32 x 32 tiles, K over four waves, K % 512 == 0), with an optional fused LayerNorm over K (a.ln_g / a.ln_b / a.ln_eps)
bool gemm3_s_takes(const DGemmArgs& a, int epi);
int launch_gemm3_s(const DGemmArgs& a, int epi, hipStream_t st);

// softmax(Q K^T * scale) V per (sequence, head) on the same arithmetic, head_dim 64.  Rows of Q / K / V / O are fp32 with arbitrary row strides;
// sequence s covers q rows [q_row0(s), + q_len(s)) and kv rows [kv_row0(s), + kv_len(s)).  Two addressing modes:
//  * strided (extractor): q_row0 = s * q_seq_rows etc., all sequences the same lengths Sq / Sk, not causal;
//  * ragged causal (decoder prefill): per-prompt (row0, len) lists, K / V rows read from the KV cache of the prompt's slot, query t sees keys 0 .. t.
struct Attn3Args {
  const float* Q; int ldq; long long q_seq;      // strided mode: element offset between sequences
  const float* K; int ldk; long long k_seq;
  const float* V; int ldv; long long v_seq;
  float* O; int ldo; long long o_seq;
  int n_seq, n_heads, Sq, Sk;
  // ragged causal mode (seq_row0 != null): Q / O rows are global row indices row0 + t; K / V = cache base of the layer: [slot][head][max_ctx][64]
  const int* seq_row0; const int* seq_len; const int* row_slot; long long slot_stride; int max_ctx; int max_len;
  float scale;                                    // 1 / sqrt(head_dim)
  int q_log2, k_log2, v_log2;                     // plane scales of the three operands
  double flops_hint;
};
int launch_attn3(const Attn3Args& a, hipStream_t st);
