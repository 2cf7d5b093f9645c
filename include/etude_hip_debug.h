/* etude_hip_debug.h -- diagnostic, measurement and test hooks of libetude_hip.so.
 *
 * NOT part of the drop-in boundary (include/etude_hip.h): nothing in etude_amd/'s serving path calls these.  They exist for
 * tests/ (taps, step logits, prompt assembly), tools/ (traces, aggressors, microbenchmarks) and the investigations recorded in
 * LABNOTES.md.  Same conventions as etude_hip.h (0 / negative ETD_E* codes, etd_last_error()). */
#ifndef ETUDE_HIP_DEBUG_H
#define ETUDE_HIP_DEBUG_H
#include "etude_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* measurement hook: device time per dependent (empty) kernel, launched eagerly vs replayed from a hipGraph */
int etd_debug_boundary_cost(int n_nodes, int iters, int big_args, void* stream, double* eager_us, double* graph_us);
/* measurement hook: average time (us) of the token-major bf16 GEMM kernel on a synthetic [M,K] x [N,K]^T problem (N % 256 == 0, K % 128 == 0) */
int etd_debug_linear(int M, int N, int K, int iters, void* stream, double* us);
/* Diagnostic aggressors (tools/probe_race.py): `iters` launches of one kernel type on private random buffers:
   which 0 = k_attn (extractor shape), 1 = k_attn causal ragged (prefill shape), 2 = k_linear with the LayerNorm epilogue, 3 = k_ln_rows. */
int etd_debug_kernel_loop(int which, int iters, void* stream);
/* Diagnostic: one launch of an empty kernel with k_embed's footprint (82 KiB static LDS, 296 registers) on the given grid. */
int etd_debug_empty_launch(int gx, int gy, int gz, int* sink_dev, void* stream);

/* test hook: fp32 velocity logits of the time heads [rows][n_note][128] for the NEXT transcript call (NULL = off) */
int etd_extractor_debug_vel_logits(etd_ext*, float* vel_logits_dev);
/* test hook: after stage s of the FIRST chunk copy the bf16 activation buffer to dst_dev (NULL = off).
 * 0 embed, 1-3 encoder layers, 4-6 freq-decoder layers [frames*n_note][256], 7 time input, 8-10 time layers. */
int etd_extractor_debug_tap(etd_ext*, int stage, void* dst_dev);

/* test hook (host only): the prompt etd_decoder_run_jobs builds for a bar given n_hist past (X, Y, attrs4) pairs; attrs4_out is [4][cap] */
int etd_debug_assemble_prompt(const etd_sched_cfg* cfg, int n_hist, const int32_t* const* hx, const int32_t* hxn, const int32_t* const* hy,
                              const int32_t* hyn, const int32_t* hattrs4, const int32_t* x, int xn, const int32_t* y_attrs4,
                              int32_t* ids_out, int32_t* cls_out, int32_t* attrs4_out, int cap, int* T_out);

/* Diagnostic (tools/probe_race.py): weighted 64-bit sums over the words of the handle's KV cache, workspaces and stream state:
   out[0] = everything, out[1 + i] = its i-th allocation (as many as `cap` allows). */
int etd_debug_decoder_checksum(etd_dec*, unsigned long long* out, int cap, void* stream);
/* Diagnostic: out[(layer * max_streams + slot) * max_ctx + pos] = 32-bit sum over the K and V rows of that position (bf16 handles). */
int etd_debug_decoder_kv_rowsums(etd_dec*, unsigned* out_host, long long cap, void* stream);
/* Diagnostic step trace (tools/probe_trace.py): after trace_begin every bf16 decode step records a hash of each row of each kernel's
 * outputs into a ring of cap_steps records of (49 * layers + 2) * n_active words; trace_read copies the ring and the step count. */
int etd_debug_decoder_trace_begin(etd_dec*, int cap_steps, void* stream);
/* layer 0's 12 split-K slabs [12][n_active][512] of the last traced step */
int etd_debug_decoder_trace_slabs(etd_dec*, float* out_host, long long cap_floats, int n_active, void* stream);
/* layer 0's queries [n_active][hidden] of the last traced step; one (layer, slot, head)'s K / V cache rows [n_pos][64] as bf16 bit patterns */
/* per-lane softmax state of layer 0's attention workgroups in the last traced step: [heads][n_active][256][8] =
 * lr, mr, o[0] after the key loop; lr after merge stages 8, 16, 32; o[0] after stages 8 and 32 */
int etd_debug_decoder_trace_lanes(etd_dec*, float* out_host, long long cap_floats, int n_active, void* stream);
int etd_debug_decoder_trace_q(etd_dec*, float* out_host, long long cap_floats, int n_active, void* stream);
int etd_debug_decoder_peek_kv(etd_dec*, int layer, int slot, int head, int n_pos, unsigned short* k_out, unsigned short* v_out, void* stream);
int etd_debug_decoder_trace_read(etd_dec*, unsigned* out_host, long long cap_words, int n_active, int* steps_done, void* stream);

/* test hook: pin the attention form of the fused bf16 decode step: -1 = the library's rule (a function of rows and mean context), 0 = one row per
 * 4-wave workgroup, 1 = two rows of a head per 8-wave workgroup.  Both forms give bit-identical results (tests/test_gpu_decoder_parity.py). */
int etd_debug_decoder_force_pair(etd_dec*, int mode);
/* test hook: switch the per-step logit store of the decode step on / off and (out_host != NULL) copy out the LAST step's logits
 * [n_active][vocab] fp32 -- the fused bf16 step keeps its logits in LDS otherwise.  Stamped / logged steps use their own captured graphs. */
int etd_debug_decoder_step_logits(etd_dec*, int on, float* out_host, int n_active, void* stream);

/* test hooks of the fp32-grade f16-split kernels (csrc/gemm3.h; tests/test_gpu_gemm3.py):
 * y[M][N] = x[M][K] w[N][K]^T + bias (x, y device fp32 row-major; w, bias host; x_bound = bound of |x| for the plane scale; gelu != 0: erf-GELU epilogue).
 * 2 .. 512 rows with K % 512 == 0 take the weight-streaming kernel (k_gemm3_s), which can apply LayerNorm(x; ln_g, ln_b, eps 1e-5) over K first (host vectors or NULL;
 * x_bound then bounds the LayerNorm output); everything else the 128 x 128 tile kernel (k_gemm3, no fused LayerNorm). */
int etd_debug_gemm3(const float* x_dev, int M, int K, const float* w_host, const float* bias_host, int N, float x_bound, int gelu, float* y_dev,
                    const float* ln_g_host, const float* ln_b_host, void* stream);
/* o = softmax(q k^T / 8) v per (sequence, head), head_dim 64: q / o [n_seq][Sq][heads * 64], k / v [n_seq][Sk][heads * 64] device fp32; causal != 0: query t sees keys 0 .. t
 * through the RAGGED path (K / V then laid out as a KV cache [n_seq slots][heads][Sk][64], lens_host[n_seq] prompt lengths <= Sq == Sk, q / o rows packed prompt after prompt) */
int etd_debug_attn3(const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int n_seq, int n_heads, int Sq, int Sk, float q_bound, float k_bound, float v_bound,
                    int causal, const int32_t* lens_host, void* stream);

/* host-only test hook: the load-time bounds behind the plane scales of csrc/gemm3.h -- out4 = { bound of LayerNorm(.; g, b) over K features, bound of W LN(.) + c,
 * bound of W x + c for |x| <= elem_bound, largest |value| in the packed f16 planes of W }, log2_out4 = the scale logarithms chosen for the three bounds and for W */
int etd_debug_g3_bounds(const float* W, const float* c, int N, int K, const float* g, const float* b, float elem_bound, float* out4, int32_t* log2_out4);

#ifdef __cplusplus
}
#endif
#endif
