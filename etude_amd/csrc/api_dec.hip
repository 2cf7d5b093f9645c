// C-ABI glue for the Decode stage: EtudeDecoder weights -> device layout, per-stream KV cache and
// generation state kept on the device, prefill / decode-step launch sequences.
// Reference: etude/models/etude_decoder.py:148-206 (forward), :291-343 (token loop);
// etude/utils/model_loader.py:12-60 (checkpoint contract).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/etude_hip_debug.h"
#include "dec_kernels.h"
#include "gemm3.h"
#include "ext_kernels.h"
#include "prof.h"

namespace {

// fp32 -> the decoder's 16-bit operand type (dec_kernels.h: IEEE half by default, bf16 under -DETD_DEC_BF16), round to nearest even
#if ETD_DEC_IS_F16
inline uint16_t f2bf_h(float f) { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
#else
inline uint16_t f2bf_h(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#endif

struct Lin { void* W = nullptr; float* b = nullptr; int N = 0, Npad = 0, K = 0;
             void* Wf = nullptr;      // d16 weights: a second copy in MFMA-fragment order for k_linear (the batched prefill); W stays row-major for the step kernels
             void* Wp = nullptr; int w_log2 = 0; };   // fp32 weights: again as hi / lo f16 planes for k_gemm3 (csrc/gemm3.h; W stays fp32 for the weight-streaming kernels below 513 rows)
struct Layer { float *ln1g, *ln1b, *ln2g, *ln2b; Lin qkv, dense, up, down;
               void* dense_hw = nullptr;   // d16 [heads][H][64]: attention.dense regrouped per head for k_dstep_attn_down
               Lin cat;      // decode step: [dense_4h_to_h | attention.dense] along K, so mlp + attn come out of ONE GEMM
               void* mlp_frag = nullptr;   // batched prefill: up | (down | dense) as k_dmlp_fused's weight stream (H 512, I 2048)
               // fp32 mode on the f16 matrix cores: log2 of the plane scales of every GEMM / attention operand, from provable bounds (etd_decoder_create)
               int x1_log2 = 0, x2_log2 = 0, q_log2 = 0, k_log2 = 0, v_log2 = 0, m_log2 = 0; };

}  // namespace

#define ETD_STAMP_WORDS (ETD_STAMP_HDR + 2 * ETD_STAMP_LOGCAP)      // u64 words of the device-side span accumulator + launch log of the attention launches (layout: dec_kernels.h)
struct etd_dec {
  etd_dec_cfg cfg;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;               // parallel to allocs (etd_debug_decoder_checksum)
  bool bf16w = false;
  int H, I, V, L, nh, S, ctx, Mmax, Mcap, out_cap;
  float *word = nullptr, *cls_emb = nullptr, *attr_tab = nullptr;
  std::vector<Layer> layers;
  float *lnfg = nullptr, *lnfb = nullptr;
  Lin head; int xf_log2 = 0;                     // (fp32 mode: plane scale of final_layer_norm's output)
  float *X1f = nullptr, *X2f = nullptr;          // fp32 mode, >= G3_MIN_ROWS rows: the two LayerNorm branches as fp32 rows
  float *t_q = nullptr, *t_ao = nullptr, *t_do = nullptr, *t_m1 = nullptr;   // fp32 mode: the last layer's tail on the prompts' last rows only ([S][H], [S][I])
  void* head_frag = nullptr;                     // lm_head in MFMA-fragment order for k_dstep_head: [tile][k-step][lane][8] (d16 weights, H == 512)
  float *rope_cos = nullptr, *rope_sin = nullptr;
  void *Kc = nullptr, *Vc = nullptr;             // [layer][slot][head][ctx][64]
  long long slot_stride = 0, layer_stride = 0;   // elements
  // workspaces
  float *h = nullptr, *h2 = nullptr, *Q = nullptr, *AO = nullptr, *DO = nullptr, *M1 = nullptr, *logits = nullptr;
  int *row_slot = nullptr, *row_pos = nullptr, *row_active = nullptr, *row_sp = nullptr, *ids = nullptr, *slots_dev = nullptr;
  // stream state
  int *cur_tok = nullptr, *len = nullptr, *done = nullptr, *n_out = nullptr, *eos = nullptr, *limit = nullptr, *tgt_attrs = nullptr, *out_tok = nullptr;
  float* tgt_proj = nullptr;                     // [slot][H]: attribute projection of the slot's target attributes (k_slot_proj)
  std::vector<int> last_slots;                   // host copy of what slots_dev holds
  float* qkv_raw = nullptr;                      // [3H] scratch row of the M == 1 QKV path
  d16 *X1b = nullptr, *X2b = nullptr, *AOb = nullptr, *M1b = nullptr;   // d16 activations of the d16 pipeline (M > 1)
  d16* Qb = nullptr;                            // batched prefill: RoPE'd queries [M][H] of the MFMA attention (k_pattn reads K / V from the cache)
  float* hlast = nullptr;                        // [S][H] gathered last rows of a batched prefill
  DSampleCfg* samp_dev = nullptr;                // sampling parameters (device-resident: captured graphs follow set_sampling)
  unsigned long long* rng_key = nullptr;         // [S] per-stream draw keys
  std::vector<unsigned long long> host_key; bool keys_dirty = true; bool sampling = false;
  float* Pk = nullptr;                           // [5][512][H] split-K partials of the decode-step (down | dense) projection
  // diagnostic (etd_debug_decoder_trace_*): per-row hashes of every decode-step kernel's outputs, one record per step
  unsigned* trace = nullptr; int* trace_step = nullptr; int trace_cap = 0; float* trace_pk = nullptr; float* trace_q = nullptr; float* trace_dbg = nullptr;   // trace_pk: layer 0's slabs of the LAST traced step
  int* row_cnt = nullptr;                        // [L][512] arrival counters of the in-launch row finish (DRowFin); zero between launches
  d16* Xcat = nullptr;                          // [512][I + H] d16: GELU(up) | attention output, the K-concatenated input of that GEMM
  std::vector<int> stage;                        // host staging of a prefill batch (fallback when the pinned buffer is absent)
  // pinned host memory (hipHostMalloc): copies to / from it are true async DMAs -- a pageable source or destination costs a
  // staging pass and ~100 us per call at these sizes, all of it with this engine's queue empty (bar boundaries)
  int* pin_stage = nullptr; size_t pin_stage_ints = 0; hipEvent_t pin_stage_evt = nullptr;   // prefill upload; the event = "the last upload has left the buffer"
  int* pin_rb = nullptr;                         // read-back: [done S][n_out S][tokens S * out_cap]
  std::vector<int> host_n_out; bool host_n_out_valid = false;   // n_out as of the last poll; valid until the next step / begin_bars
  std::map<int, hipGraphExec_t> graphs;          // captured decode step per (n_active, paired rows, rows_identity): key 4 * n_active + 2 * pair + identity
  bool rows_identity = false;                    // the step's slot list is 0, 1, ..., n_active - 1
  std::vector<int> host_len;                     // host-side estimate of each slot's KV length (profiler byte counts only)
  double attn_bytes_hint = 0;
  bool step_pair = false;                        // this call's decode steps pair the rows of a head in the attention launch (etd_decoder_step decides)
  // counters of the decode steps issued on this handle since the last etd_decoder_stats_reset (host-side, exact: etd_decoder_stats)
  double stat_steps = 0, stat_row_steps = 0, stat_kv_bytes = 0, stat_attn_launches = 0, stat_stamp_bytes = 0;
  int force_pair = -1;                           // test hook (etd_debug_decoder_force_pair): -1 = the rule above, 0 / 1 = one-row / paired-rows attention form
  unsigned long long* stamp_dev = nullptr;       // device-side span accumulator of k_dstep_attn_down (etd_decoder_stamp); its own allocation
  bool stamp_on = false, stamp_armed = false; long long stamp_skip = 0;      // armed: requested; on: this call's steps are stamped (after `stamp_skip` more steps)
  float* logits_dbg = nullptr; bool logits_dbg_on = false, last_step_fused = false;   // test hook: the fused step's logits [S][V] (etd_debug_decoder_step_logits)
  // weight sharing (etd_decoder_clone): a clone reads the owner's weight buffers and has its own KV cache, workspaces and
  // stream state.  `allocs` of an owner = weights first (n_weight_allocs of them), then its workspaces; a clone's = workspaces only.
  etd_dec* weights_owner = nullptr;              // null: this handle owns its weights
  size_t n_weight_allocs = 0;
  int n_clones = 0; bool zombie = false;         // owner destroyed while clones are alive: weights freed with the last clone

  template <typename T> int alloc(T** p, size_t n, bool zero = false) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, n * sizeof(T) + 256));
    if (zero) HIP_TRY(hipMemset(q, 0, n * sizeof(T) + 256));
    allocs.push_back(q);
    alloc_bytes.push_back(n * sizeof(T) + 256);
    *p = (T*)q;
    return ETD_OK;
  }
};

namespace {

struct Loader {
  std::map<std::string, std::pair<const float*, int64_t>> t;
  const float* get(const std::string& k, int64_t numel) {
    auto it = t.find(k);
    if (it == t.end()) { g_etd_err = "missing weight '" + k + "'"; return nullptr; }
    if (it->second.second != numel) { g_etd_err = "weight '" + k + "' has " + std::to_string(it->second.second) + " elements, expected " + std::to_string(numel); return nullptr; }
    return it->second.first;
  }
};

int up_f32(etd_dec* d, float** dst, const float* src, size_t n) {
  ETD_TRY(d->alloc(dst, n));
  HIP_TRY(hipMemcpy(*dst, src, n * 4, hipMemcpyHostToDevice));
  return ETD_OK;
}

// weight [N][K] (+ bias [N] or null) -> device, rows padded with zeros to a multiple of 128
int load_lin(etd_dec* d, Loader& L, const std::string& pfx, int N, int K, bool has_bias, Lin* w) {
  const float* W = L.get(pfx + ".weight", (int64_t)N * K);
  if (!W) return ETD_EINVAL;
  const float* b = nullptr;
  if (has_bias) { b = L.get(pfx + ".bias", N); if (!b) return ETD_EINVAL; }
  const int Npad = ((N + 127) / 128) * 128;
  w->N = N; w->Npad = Npad; w->K = K;
  if (d->bf16w) {
    std::vector<uint16_t> hb((size_t)Npad * K, 0);
    for (size_t i = 0; i < (size_t)N * K; ++i) hb[i] = f2bf_h(W[i]);
    uint16_t* p; ETD_TRY(d->alloc(&p, hb.size()));
    HIP_TRY(hipMemcpy(p, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    w->W = p;
    if (K % 16 == 0) {
      std::vector<uint16_t> hp(hb.size());
      pack_wfrag_host(hb.data(), Npad, K, hp.data());
      uint16_t* pf; ETD_TRY(d->alloc(&pf, hp.size()));
      HIP_TRY(hipMemcpy(pf, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
      w->Wf = pf;
    }
  } else {
    std::vector<float> hf((size_t)Npad * K, 0.f);
    memcpy(hf.data(), W, (size_t)N * K * 4);
    float* p; ETD_TRY(d->alloc(&p, hf.size()));
    HIP_TRY(hipMemcpy(p, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    w->W = p;
    if (K % 32 == 0) {
      std::vector<uint16_t> planes(g3_packed_elems(Npad, K));
      w->w_log2 = g3_pack_weights_host(W, N, Npad, K, planes.data());
      uint16_t* pp; ETD_TRY(d->alloc(&pp, planes.size()));
      HIP_TRY(hipMemcpy(pp, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
      w->Wp = pp;
    }
  }
  std::vector<float> hb2(Npad, 0.f);
  if (b) memcpy(hb2.data(), b, (size_t)N * 4);
  ETD_TRY(up_f32(d, &w->b, hb2.data(), Npad));
  return ETD_OK;
}

int load_vec(etd_dec* d, Loader& L, const std::string& name, int n, float** dst) {
  const float* p = L.get(name, n);
  if (!p) return ETD_EINVAL;
  return up_f32(d, dst, p, n);
}

// ---- one forward pass over M rows.  `rows` describes each row (slot, position, active); the embeddings are in
// d->h.  Returns (in *hfinal) the buffer that holds the last layer's output (before the final LayerNorm).
//
// fp32 weights, or M == 1: fp32 activations, LayerNorm fused into the GEMM prologues (k_dgemm / k_dgemm_s / k_dgemv).
// d16 weights, M > 1:     k_ln_rows -> d16 activations -> big-tile MFMA GEMM (k_linear decoder modes, M > 512, the
//                          batched prefill) or the K-split skinny GEMM (M <= 512, the batched decode step).
struct PrefillInfo { int n; const int* seq_row0; const int* seq_len; int max_len; double attn_flops; };
// "Only each prompt's last position is needed" (begin_bars): the last layer then runs its attention, MLP and residual for
// those n rows only -- on the decode-step kernels -- after the big QKV GEMM has put every position's K/V into the cache.
// ---- step trace (diagnostic): hash of every row of a buffer -> trace[step % cap][off + slab * M + row]
__global__ __launch_bounds__(64) void k_trace_rows(const unsigned* buf, long long row_stride, int words, long long slab_stride, int M,
                                                   unsigned* trace, const int* step, int cap, long long wps, int off) {
  const int row = blockIdx.x, slab = blockIdx.y, lane = threadIdx.x;
  const unsigned* p = buf + (long long)slab * slab_stride + (long long)row * row_stride;
  unsigned acc = 0;
  for (int i = lane; i < words; i += 64) acc += p[i] * (2654435761u * (unsigned)(i + 1) | 1u);
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) trace[(long long)(*step % cap) * wps + off + slab * M + row] = acc;
}
// K / V cache of (row, head) as the step's attention will read it: hash of the row appended by this step, and of rows 0 .. pos
__global__ __launch_bounds__(64) void k_trace_kv(const unsigned* Kc, const unsigned* Vc, long long slot_stride_w, int max_ctx, const int* row_slot, const int* row_pos, int M,
                                                 unsigned* trace, const int* step, int cap, long long wps, int off) {
  const int row = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const int slot = row_slot[row], pos = row_pos[row] < max_ctx ? row_pos[row] : max_ctx - 1;
  unsigned* out = trace + (long long)(*step % cap) * wps + off;
  for (int which = 0; which < 2; ++which) {
    const unsigned* base = (which ? Vc : Kc) + (long long)slot * slot_stride_w + (long long)head * max_ctx * 32;      // 64 d16 = 32 words per position
    unsigned a_new = 0, a_all = 0;
    for (long long i = lane; i < (long long)(pos + 1) * 32; i += 64) {
      const unsigned w = base[i] * (2654435761u * (unsigned)(i + 1) | 1u);
      a_all += w;
      if (i >= (long long)pos * 32) a_new += w;
    }
    for (int o = 32; o; o >>= 1) { a_new += __shfl_xor(a_new, o); a_all += __shfl_xor(a_all, o); }
    if (lane == 0) { out[(which * 8 + head) * M + row] = a_new; out[(16 + which * 8 + head) * M + row] = a_all; }
  }
}
__global__ void k_trace_next(int* step) { *step += 1; }
#define ETD_TRACE_LAYER 49     // records per row and layer: Q, gelu(up), 12 slabs, h_out, ln1, ln2, then per head K[pos], V[pos], K[0..pos], V[0..pos]
static inline long long trace_wps(const etd_dec* d, int M) { return (long long)(ETD_TRACE_LAYER * d->L + 2) * M; }
static int trace_rows(etd_dec* d, const void* buf, long long row_stride, int words, int nslab, long long slab_stride, int M, int off, hipStream_t st) {
  hipLaunchKernelGGL(k_trace_rows, dim3(M, nslab), dim3(64), 0, st, (const unsigned*)buf, row_stride, words, slab_stride, M, d->trace, d->trace_step,
                     d->trace_cap, trace_wps(d, M), off);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}

static inline bool fused_pmlp_on() { const char* e = getenv("ETD_FUSED_PMLP"); return !e || atoi(e) > 0; }
struct LastOnly { int n; const int* idx; DecRows rows; };

static DGemmArgs g3_args(const float* X, int ldx, const Lin& w, int x_log2, int M) {
  DGemmArgs a = {};
  a.X = X; a.ldx = ldx; a.W = w.W; a.Wp = w.Wp; a.w_log2 = w.w_log2; a.x_log2 = x_log2; a.bias = w.b; a.M = M; a.N = w.N; a.Npad = w.Npad; a.K = w.K;
  return a;
}

// ---- fp32 mode, >= G3_MIN_ROWS rows: the layer as LayerNorm rows -> QKV (+RoPE, KV append) -> attention -> dense -> up + GELU -> down + residual with every
// contraction on the f16 matrix cores at fp32 grade (csrc/gemm3.h).  Prefill: ragged causal attention over the prompts straight from the fp32 cache rows the QKV
// epilogue has just written, and the last layer's tail only for each prompt's last position; decode step: k_dattn<float> per (row, head).
static int gemm3_auto(const DGemmArgs& a, int epi, hipStream_t st) { return gemm3_s_takes(a, epi) ? launch_gemm3_s(a, epi, st) : launch_gemm3(a, epi, st); }
int forward_body_x3(etd_dec* d, int M, const DecRows& rows, float** hfinal, hipStream_t st, const PrefillInfo* pf, const LastOnly* lo, bool* compact) {
  float* hin = d->h; float* hout = d->h2;
  const int H = d->H;
  for (int l = 0; l < d->L; ++l) {
    const Layer& w = d->layers[l];
    void* Kl = (char*)d->Kc + (size_t)l * d->layer_stride * 4;
    void* Vl = (char*)d->Vc + (size_t)l * d->layer_stride * 4;
    ETD_TRY(launch_ln_rows_f32(hin, M, H, w.ln1g, w.ln1b, w.ln2g, w.ln2b, d->cfg.layer_norm_eps, d->X1f, d->X2f, st));
    DGemmArgs q = g3_args(d->X1f, H, w.qkv, w.x1_log2, M);
    q.rows = rows; q.rope_cos = d->rope_cos; q.rope_sin = d->rope_sin; q.rot_half = 8; q.Q = d->Q;
    q.Kc = Kl; q.Vc = Vl; q.slot_stride = d->slot_stride; q.max_ctx = d->ctx; q.n_heads = d->nh;
    ETD_TRY(gemm3_auto(q, DEPI_QKV, st));
    if (l == d->L - 1 && lo && pf && lo->n >= 1 && lo->n < G3_MIN_ROWS && lo->n <= d->S) {
      // last layer: every position's K / V is in the cache now; attention, MLP and residual are needed for the prompts' last rows only
      const int n = lo->n;
      ETD_TRY(launch_gather_rows(hin, lo->idx, n, H, d->hlast, st));
      ETD_TRY(launch_gather_rows(d->Q, lo->idx, n, H, d->t_q, st));
      DAttnArgs at = {};
      at.Q = d->t_q; at.Kc = Kl; at.Vc = Vl; at.slot_stride = d->slot_stride; at.max_ctx = d->ctx; at.n_heads = d->nh;
      at.rows = lo->rows; at.M = n; at.O = d->t_ao; at.scale = 0.125f; at.bytes_hint = 0;
      ETD_TRY(launch_dattn(at, false, st));
      DGemmArgs de = g3_args(d->t_ao, H, w.dense, w.v_log2, n); de.Y = d->t_do; de.ldy = H;
      ETD_TRY(launch_dgemm(de, DEPI_BIAS, false, st));
      DGemmArgs up = g3_args(d->hlast, H, w.up, w.x2_log2, n); up.Y = d->t_m1; up.ldy = d->I; up.ln_g = w.ln2g; up.ln_b = w.ln2b; up.ln_eps = d->cfg.layer_norm_eps;
      ETD_TRY(launch_dgemm(up, DEPI_GELU, false, st));
      DGemmArgs dn = g3_args(d->t_m1, d->I, w.down, w.m_log2, n); dn.add = d->t_do; dn.hin = d->hlast; dn.hout = hout;
      ETD_TRY(launch_dgemm(dn, DEPI_RESID, false, st));
      *hfinal = hout;                          // rows 0 .. n-1 = the prompts' last positions, in prompt order
      if (compact) *compact = true;
      return ETD_OK;
    }
    if (pf) {
      Attn3Args t = {};
      t.Q = d->Q; t.ldq = H; t.K = (const float*)Kl; t.V = (const float*)Vl; t.O = d->AO; t.ldo = H; t.n_seq = pf->n; t.n_heads = d->nh;
      t.seq_row0 = pf->seq_row0; t.seq_len = pf->seq_len; t.row_slot = rows.slot; t.slot_stride = d->slot_stride; t.max_ctx = d->ctx; t.max_len = pf->max_len;
      t.scale = 0.125f; t.q_log2 = w.q_log2; t.k_log2 = w.k_log2; t.v_log2 = w.v_log2; t.flops_hint = pf->attn_flops;
      ETD_TRY(launch_attn3(t, st));
    } else {
      DAttnArgs at = {};
      at.Q = d->Q; at.Kc = Kl; at.Vc = Vl; at.slot_stride = d->slot_stride; at.max_ctx = d->ctx; at.n_heads = d->nh;
      at.rows = rows; at.M = M; at.O = d->AO; at.scale = 0.125f; at.bytes_hint = d->attn_bytes_hint;
      if (rows.slot == d->row_slot) { at.row_sp = d->row_sp; at.identity = d->rows_identity ? 1 : 0; }
      at.stamp = d->stamp_on ? d->stamp_dev : nullptr; at.stamp_par = l & 1;
      ETD_TRY(launch_dattn(at, false, st));
    }
    DGemmArgs de = g3_args(d->AO, H, w.dense, w.v_log2, M); de.Y = d->DO; de.ldy = H;      // (attention output: a convex combination of V rows)
    ETD_TRY(gemm3_auto(de, DEPI_BIAS, st));
    DGemmArgs up = g3_args(d->X2f, H, w.up, w.x2_log2, M); up.Y = d->M1; up.ldy = d->I;
    ETD_TRY(gemm3_auto(up, DEPI_GELU, st));
    DGemmArgs dn = g3_args(d->M1, d->I, w.down, w.m_log2, M); dn.add = d->DO; dn.hin = hin; dn.hout = hout;
    ETD_TRY(gemm3_auto(dn, DEPI_RESID, st));
    float* t = hin; hin = hout; hout = t;
  }
  *hfinal = hin;
  return ETD_OK;
}

int forward_body(etd_dec* d, int M, const DecRows& rows, float** hfinal, hipStream_t st, const PrefillInfo* pf = nullptr, bool ln0_done = false,
                 const LastOnly* lo = nullptr, bool* compact = nullptr, bool is_step = false) {
  if (compact) *compact = false;
  if (!d->bf16w && M >= G3_MIN_ROWS && d->X1f && d->layers[0].qkv.Wp && d->layers[0].down.Wp && g3_enabled()) return forward_body_x3(d, M, rows, hfinal, st, pf, lo, compact);
  float* hin = d->h; float* hout = d->h2;
  const size_t esz = d->bf16w ? 2 : 4;
  const bool bpipe = d->bf16w && (M > 1 || is_step);     // (is_step with M == 1: the fused step kernels for a single stream, ETD_FUSED_M1)
  // a decode step (is_step: one row per stream, M <= DS_STEP_MAX_ROWS) stays on the fused step kernels whatever its row count
  const bool big = bpipe && !is_step && M > DS_MAX_ROWS && d->layers[0].qkv.Wf && d->layers[0].up.Wf && d->layers[0].cat.Wf;
  bool ln_ready = false;            // X1b / X2b already hold this layer's LayerNorm rows (written by the previous layer's k_dmlp_fused)
  for (int l = 0; l < d->L; ++l) {
    const Layer& w = d->layers[l];
    void* Kl = (char*)d->Kc + (size_t)l * d->layer_stride * esz;
    void* Vl = (char*)d->Vc + (size_t)l * d->layer_stride * esz;
    const bool small = bpipe && !big && (d->I + d->H) % (5 * 64 * 8) == 0;            // decode step: split-K down projection + fused (partial-sum, residual, next LayerNorm) kernel
    const bool catk = small || big;     // attention.dense folded into the down projection: [W2 | Wd] [gelu(..) ; attn] + (b2 + bd), one GEMM and no fp32 round trip of the dense output
    // batched prefill on the fused MLP kernel: it also writes the NEXT layer's LayerNorm rows, so only layer 0 needs the row kernel
    // ON by default, ETD_FUSED_PMLP=0 turns it off (read at create time for the packed stream, and per call so that the A/B test can toggle it): measured round 2
    // ((history: 4ac2f57) tools/runs/r2_run47.sh) at 195 us per launch against 166 + 16 us for the launches it replaces, +1.3 % in the job -- csrc/dec_fused.hip
    const bool fmlp = big && w.mlp_frag && d->H == 512 && d->I == 2048 && fused_pmlp_on();
    if (bpipe && (!small || (l == 0 && !ln0_done)) && !ln_ready) ETD_TRY(launch_ln_rows(hin, M, d->H, w.ln1g, w.ln1b, w.ln2g, w.ln2b, d->cfg.layer_norm_eps, d->X1b, d->X2b, st));
    ln_ready = false;
    // ---- fused QKV + RoPE + KV append
    DGemmArgs q = {};
    q.X = hin; q.ldx = d->H; q.W = w.qkv.W; q.Wf = w.qkv.Wf; q.bias = w.qkv.b; q.M = M; q.N = w.qkv.N; q.Npad = w.qkv.Npad; q.K = d->H;
    q.Wp = w.qkv.Wp; q.w_log2 = w.qkv.w_log2; q.x_log2 = w.x1_log2;
    if (bpipe) q.Xb = d->X1b; else { q.ln_g = w.ln1g; q.ln_b = w.ln1b; q.ln_eps = d->cfg.layer_norm_eps; }
    q.Y = d->qkv_raw; q.ldy = 3 * d->H;
    q.rows = rows; q.rope_cos = d->rope_cos; q.rope_sin = d->rope_sin; q.rot_half = 8; q.Q = d->Q;
    q.Kc = Kl; q.Vc = Vl; q.slot_stride = d->slot_stride; q.max_ctx = d->ctx; q.n_heads = d->nh;
    const bool mfma_attn = big && pf != nullptr;
    if (mfma_attn) q.Qb = d->Qb;       // (K / V: the attention reads the cache rows this epilogue writes)
    if (big) {
      static const bool pqkv_on = !getenv("ETD_NO_PQKV");
      if (mfma_attn && pqkv_on && M >= 49152 && d->H == 512 && d->nh == 8 && w.qkv.N == 1536) {
        // QKV with the token block stationary in registers and the weights streamed through LDS (csrc/dec_prefill.hip): 256-token workgroups that walk all 48 weight
        // tiles -- from ~192 workgroups up (below that k_linear's 128 x 256 tiles fill the chip better: 6 466 rows 24 us against 86)
        PQkvArgs pa = {};
        pa.X = d->X1b; pa.ldx = d->H; pa.Wf = (const d16*)w.qkv.Wf; pa.bias = w.qkv.b; pa.M = M; pa.N = w.qkv.N; pa.rows = rows;
        pa.rope_cos = d->rope_cos; pa.rope_sin = d->rope_sin; pa.Qb = d->Qb; pa.Kc = (d16*)Kl; pa.Vc = (d16*)Vl; pa.slot_stride = d->slot_stride;
        pa.max_ctx = d->ctx; pa.n_heads = d->nh;
        ETD_TRY(launch_pqkv(pa, st));
      } else {
        LinArgs a = {};
        a.X = (const e16*)d->X1b; a.ldx = d->H; a.W = (const e16*)w.qkv.Wf;     /* (LinArgs carries the extractor's element type; the decoder modes of k_linear read these as d16) */ a.bias = w.qkv.b; a.M = M; a.N = w.qkv.N; a.K = d->H; a.vt_block = -1; a.dec = q;
        ETD_TRY(launch_linear_dec(a, DEPI_QKV, st));
      }
      if (l == d->L - 1 && lo && mfma_attn && lo->n > 1 && lo->n <= DS_STEP_MAX_ROWS && d->H == 512 && (d->I + d->H) % (5 * 64 * 8) == 0 && !getenv("ETD_NO_LAST_ONLY")) {
        // last layer, last positions only: every position's K/V is in the cache now; what remains of the layer is needed for
        // n rows, not M (attention, MLP up, (down | dense), residual = 9 % of the prefill's FLOPs at 8 layers)
        const int n = lo->n;
        ETD_TRY(launch_gather_rows(hin, lo->idx, n, d->H, d->hlast, st));
        ETD_TRY(launch_ln_rows(d->hlast, n, d->H, w.ln1g, w.ln1b, w.ln2g, w.ln2b, d->cfg.layer_norm_eps, d->X1b, d->X2b, st));
        DGemmArgs q2 = q;
        q2.X = d->hlast; q2.M = n; q2.rows = lo->rows; q2.Xb = d->X1b; q2.Qb = nullptr;
        DGemmArgs up2 = {};
        up2.X = d->hlast; up2.ldx = d->H; up2.W = w.up.W; up2.Wf = w.up.Wf; up2.bias = w.up.b; up2.M = n; up2.N = d->I; up2.Npad = w.up.Npad; up2.K = d->H;
        up2.Y = d->M1; up2.Xb = d->X2b; up2.Yb = d->Xcat; up2.ldy = d->I + d->H;
        ETD_TRY(launch_dstep_qkv_up(q2, up2, st));
        DAttnArgs at = {};
        at.Q = d->Q; at.Kc = Kl; at.Vc = Vl; at.slot_stride = d->slot_stride; at.max_ctx = d->ctx; at.n_heads = d->nh;
        at.rows = lo->rows; at.M = n; at.O = d->AO; at.Ob = d->Xcat + d->I; at.ldob = d->I + d->H; at.scale = 0.125f; at.bytes_hint = 0;
        ETD_TRY(launch_dattn(at, d->bf16w, st));
        DGemmArgs dn2 = {};
        dn2.X = d->M1; dn2.Xb = d->Xcat; dn2.ldx = d->I + d->H; dn2.W = w.cat.W; dn2.bias = w.cat.b; dn2.M = n; dn2.N = d->H; dn2.Npad = w.cat.Npad; dn2.K = d->I + d->H;
        dn2.hin = d->hlast; dn2.hout = hout; dn2.k_splits = 5; dn2.Y = d->Pk; dn2.ldy = d->H;
        ETD_TRY(launch_dgemm(dn2, DEPI_PARTIAL, true, st));
        ETD_TRY(launch_resid_ln_rows(d->Pk, 5, w.cat.b, nullptr, d->hlast, hout, n, d->H, nullptr, nullptr, nullptr, nullptr, d->cfg.layer_norm_eps, nullptr, nullptr, st));
        *hfinal = hout;                        // rows 0 .. n-1 = the prompts' last positions, in prompt order
        if (compact) *compact = true;
        return ETD_OK;
      }
    } else if (small) {
      // decode step: QKV (+RoPE, KV append) and MLP up (+GELU -> Xcat) share one launch
      DGemmArgs up = {};
      up.X = hin; up.ldx = d->H; up.W = w.up.W; up.Wf = w.up.Wf; up.bias = w.up.b; up.M = M; up.N = d->I; up.Npad = w.up.Npad; up.K = d->H;
      up.Y = d->M1; up.Xb = d->X2b; up.Yb = d->Xcat; up.ldy = d->I + d->H;
      ETD_TRY(launch_dstep_qkv_up(q, up, st));
      if (d->trace) {
        ETD_TRY(trace_rows(d, d->Q, d->H, d->H, 1, 0, M, l * ETD_TRACE_LAYER * M, st));
        if (l == 0) HIP_TRY(hipMemcpyAsync(d->trace_q, d->Q, (size_t)M * d->H * 4, hipMemcpyDeviceToDevice, st));
        ETD_TRY(trace_rows(d, d->Xcat, (d->I + d->H) / 2, d->I / 2, 1, 0, M, l * ETD_TRACE_LAYER * M + M, st));
        if (d->nh == 8) {
          hipLaunchKernelGGL(k_trace_kv, dim3(M, 8), dim3(64), 0, st, (const unsigned*)Kl, (const unsigned*)Vl, d->slot_stride / 2, d->ctx, d->row_slot, d->row_pos, M,
                             d->trace, d->trace_step, d->trace_cap, trace_wps(d, M), l * ETD_TRACE_LAYER * M + 17 * M);
          HIP_TRY(hipGetLastError());
        }
      }
    } else {
      ETD_TRY(launch_dgemm(q, DEPI_QKV, d->bf16w, st));
    }
    // ---- decode step: attention (+ its share of attention.dense) and the MLP down projection in ONE launch, then the row kernel
    const bool attn_down = small && rows.slot == d->row_slot && d->H == 512 && d->nh == 8 && d->I % 512 == 0 && d->I / 512 + d->nh <= 12 && w.dense_hw &&
                           d->ctx >= 256 && !getenv("ETD_NO_ATTN_DOWN");
    if (attn_down) {
      DAttnArgs at = {};
      at.Q = d->Q; at.Kc = Kl; at.Vc = Vl; at.slot_stride = d->slot_stride; at.max_ctx = d->ctx; at.n_heads = d->nh;
      at.rows = rows; at.M = M; at.scale = 0.125f; at.bytes_hint = d->attn_bytes_hint;
      at.pair = d->step_pair ? 1 : 0;
      at.stamp = d->stamp_on ? d->stamp_dev : nullptr; at.stamp_par = l & 1;
      at.row_sp = d->row_sp; at.identity = d->rows_identity ? 1 : 0;
      const int ksd = d->I / 512;
      at.dense_w = (const d16*)w.dense_hw; at.dense_out = d->Pk + (size_t)ksd * M * d->H;
      at.dbg = (d->trace && l == 0) ? d->trace_dbg : nullptr;
      DGemmArgs dn = {};
      dn.Xb = d->Xcat; dn.ldx = d->I + d->H; dn.W = w.cat.W; dn.K = d->I + d->H; dn.M = M; dn.N = d->H; dn.Npad = d->H;
      dn.k_splits = ksd; dn.Y = d->Pk; dn.ldy = d->H;
      const Layer* nx = l + 1 < d->L ? &d->layers[l + 1] : nullptr;
      // ETD_ROWFIN=1: the row kernel (split-K sum + bias + residual + next LayerNorms) rides in the attention launch -- the last
      // contributor of a row finishes it (DRowFin; 17 launches per step instead of 25, bit-identical results).  Measured round 2
      // ((history: 4ac2f57) tools/runs/r2_run21/24/27.sh): a step of one engine 0.197 -> 0.185 ms, four engines stepping 9.9 -> 10.1 engine-steps/ms,
      // but the JOB 569-575 -> 567 audio-s/s (the attention workgroups live 19 instead of 15.5 us and hold 128 registers per
      // wave while the other engines' prefill GEMMs want the same CUs).  Off by default.
#ifdef ETD_EXPERIMENTS
      static const bool rowfin = getenv("ETD_ROWFIN") && atoi(getenv("ETD_ROWFIN")) > 0;
#else
      constexpr bool rowfin = false;          // (measured dead end: built with -DETD_EXPERIMENTS only)
#endif
      if (rowfin && ksd + d->nh == 12 && M <= DS_STEP_MAX_ROWS) {
        DRowFin fin = {};
        fin.cnt = d->row_cnt + (size_t)l * DS_STEP_MAX_ROWS; fin.target = d->nh + 16 * ksd; fin.P = d->Pk; fin.nslab = 12;
        fin.bias = w.cat.b; fin.hin = hin; fin.hout = hout; fin.eps = d->cfg.layer_norm_eps;
        if (nx) { fin.g1 = nx->ln1g; fin.b1 = nx->ln1b; fin.g2 = nx->ln2g; fin.b2 = nx->ln2b; fin.x1 = d->X1b; fin.x2 = d->X2b; }
        ETD_TRY(launch_dstep_attn_down(at, dn, &fin, st));
      } else {
        ETD_TRY(launch_dstep_attn_down(at, dn, nullptr, st));
        if (d->trace && ksd + d->nh == 12) ETD_TRY(trace_rows(d, d->Pk, d->H, d->H, 12, (long long)M * d->H, M, l * ETD_TRACE_LAYER * M + 2 * M, st));
        if (d->trace && l == 0 && ksd + d->nh == 12) HIP_TRY(hipMemcpyAsync(d->trace_pk, d->Pk, (size_t)12 * M * d->H * 4, hipMemcpyDeviceToDevice, st));
        ETD_TRY(launch_resid_ln_rows(d->Pk, ksd + d->nh, w.cat.b, nullptr, hin, hout, M, d->H, nx ? nx->ln1g : nullptr, nx ? nx->ln1b : nullptr,
                                     nx ? nx->ln2g : nullptr, nx ? nx->ln2b : nullptr, d->cfg.layer_norm_eps, nx ? d->X1b : nullptr, nx ? d->X2b : nullptr, st));
        if (d->trace) {
          ETD_TRY(trace_rows(d, hout, d->H, d->H, 1, 0, M, l * ETD_TRACE_LAYER * M + 14 * M, st));
          if (nx) {
            ETD_TRY(trace_rows(d, d->X1b, d->H / 2, d->H / 2, 1, 0, M, l * ETD_TRACE_LAYER * M + 15 * M, st));
            ETD_TRY(trace_rows(d, d->X2b, d->H / 2, d->H / 2, 1, 0, M, l * ETD_TRACE_LAYER * M + 16 * M, st));
          }
        }
      }
      float* t = hin; hin = hout; hout = t;
      continue;
    }
    // ---- causal attention against the slot's KV cache
    if (mfma_attn) {
      // ragged causal MFMA flash attention over the prompts of all streams at once, K / V straight from the cache rows (csrc/dec_prefill.hip)
      PAttnArgs t = {};
      t.Q = d->Qb; t.ldq = d->H; t.Kc = (const d16*)Kl; t.Vc = (const d16*)Vl; t.slot_stride = d->slot_stride; t.max_ctx = d->ctx; t.n_heads = d->nh;
      t.O = d->Xcat + d->I; t.ldo = d->I + d->H; t.seq_row0 = pf->seq_row0; t.seq_len = pf->seq_len; t.row_slot = rows.slot;
      t.n_seq = pf->n; t.max_len = pf->max_len; t.scale_log2e = 0.125f * 1.4426950408889634f; t.flops_hint = pf->attn_flops;
      ETD_TRY(launch_pattn(t, st));
    } else {
      DAttnArgs at = {};
      at.Q = d->Q; at.Kc = Kl; at.Vc = Vl; at.slot_stride = d->slot_stride; at.max_ctx = d->ctx; at.n_heads = d->nh;
      at.rows = rows; at.M = M; at.O = d->AO;
      at.Ob = bpipe ? d->AOb : nullptr; at.scale = 0.125f; at.bytes_hint = d->attn_bytes_hint;
      if (rows.slot == d->row_slot) { at.row_sp = d->row_sp; at.identity = d->rows_identity ? 1 : 0; at.stamp = d->stamp_on ? d->stamp_dev : nullptr; at.stamp_par = l & 1; }     // decode step: the step's (slot, pos) pairs
      if (catk) { at.Ob = d->Xcat + d->I; at.ldob = d->I + d->H; }
      ETD_TRY(launch_dattn(at, d->bf16w, st));
    }
    // ---- attention.dense
    DGemmArgs de = {};
    de.X = d->AO; de.ldx = d->H; de.W = w.dense.W; de.bias = w.dense.b; de.M = M; de.N = d->H; de.Npad = w.dense.Npad; de.K = d->H;
    de.Y = d->DO; de.ldy = d->H; de.Wp = w.dense.Wp; de.w_log2 = w.dense.w_log2; de.x_log2 = w.v_log2;
    if (bpipe) de.Xb = d->AOb;
    if (!catk) {                // (otherwise attention.dense is folded into the (down | dense) GEMM below)
      ETD_TRY(launch_dgemm(de, DEPI_BIAS, d->bf16w, st));
    }
    // ---- MLP up + GELU
    DGemmArgs up = {};
    up.X = hin; up.ldx = d->H; up.W = w.up.W; up.bias = w.up.b; up.M = M; up.N = d->I; up.Npad = w.up.Npad; up.K = d->H;
    up.Y = d->M1; up.ldy = d->I; up.Wp = w.up.Wp; up.w_log2 = w.up.w_log2; up.x_log2 = w.x2_log2;
    if (bpipe) { up.Xb = d->X2b; up.Yb = d->M1b; if (catk) { up.Yb = d->Xcat; up.ldy = d->I + d->H; } } else { up.ln_g = w.ln2g; up.ln_b = w.ln2b; up.ln_eps = d->cfg.layer_norm_eps; }
    if (fmlp) {
      // up + GELU, (down | dense), residual and the next layer's LayerNorms: one launch, the hidden layer never leaves the CU
      const Layer* nx = l + 1 < d->L ? &d->layers[l + 1] : nullptr;
      DMlpArgs ma = {};
      ma.X2 = d->X2b; ma.AO = d->Xcat + d->I; ma.ldao = d->I + d->H; ma.hin = hin; ma.hout = hout; ma.Wm = (const d16*)w.mlp_frag;
      ma.b_up = w.up.b; ma.b_cat = w.cat.b; ma.eps = d->cfg.layer_norm_eps; ma.M = M;
      if (nx) { ma.g1 = nx->ln1g; ma.b1 = nx->ln1b; ma.g2 = nx->ln2g; ma.b2 = nx->ln2b; ma.nx1 = d->X1b; ma.nx2 = d->X2b; }
      ETD_TRY(launch_dmlp_fused(ma, st));
      ln_ready = nx != nullptr;
      float* t = hin; hin = hout; hout = t;
      continue;
    }
    if (big) {
      LinArgs a = {};
      a.X = (const e16*)d->X2b; a.ldx = d->H; a.W = (const e16*)w.up.Wf; a.bias = w.up.b; a.M = M; a.N = d->I; a.K = d->H; a.vt_block = -1; a.dec = up;
      ETD_TRY(launch_linear_dec(a, DEPI_GELU, st));
    } else if (!small) {        // (decode step: already issued together with QKV)
      ETD_TRY(launch_dgemm(up, DEPI_GELU, d->bf16w, st));
    }
    // ---- MLP down + parallel residual: h = (mlp + attn) + h   (modeling_gpt_neox.py:272)
    DGemmArgs dn = {};
    dn.X = d->M1; dn.ldx = d->I; dn.W = w.down.W; dn.bias = w.down.b; dn.M = M; dn.N = d->H; dn.Npad = w.down.Npad; dn.K = d->I;
    dn.add = d->DO; dn.hin = hin; dn.hout = hout; dn.Wp = w.down.Wp; dn.w_log2 = w.down.w_log2; dn.x_log2 = w.m_log2;
    if (bpipe) dn.Xb = d->M1b;
    if (big) {
      dn.add = nullptr;
      LinArgs a = {};
      a.X = (const e16*)d->Xcat; a.ldx = d->I + d->H; a.W = (const e16*)w.cat.Wf; a.bias = w.cat.b; a.M = M; a.N = d->H; a.K = d->I + d->H; a.vt_block = -1; a.dec = dn;
      ETD_TRY(launch_linear_dec(a, DEPI_RESID, st));
    } else if (small) {
      // (down | dense) projection with K split over workgroups, then ONE row kernel: partial sums + bias + residual and the
      // next layer's two LayerNorms.  (Folding that row kernel into the GEMM's last-arriving workgroup was measured: the
      // serial read of 5 slabs x 32 rows costs 3x the kernel boundary it saves.)
      dn.Xb = d->Xcat; dn.ldx = d->I + d->H; dn.W = w.cat.W; dn.K = d->I + d->H; dn.Npad = w.cat.Npad;
      dn.k_splits = 5; dn.Y = d->Pk; dn.ldy = d->H;
      ETD_TRY(launch_dgemm(dn, DEPI_PARTIAL, true, st));
      const Layer* nx = l + 1 < d->L ? &d->layers[l + 1] : nullptr;
      ETD_TRY(launch_resid_ln_rows(d->Pk, 5, w.cat.b, nullptr, hin, hout, M, d->H, nx ? nx->ln1g : nullptr, nx ? nx->ln1b : nullptr,
                                   nx ? nx->ln2g : nullptr, nx ? nx->ln2b : nullptr, d->cfg.layer_norm_eps, nx ? d->X1b : nullptr, nx ? d->X2b : nullptr, st));
    } else {
      ETD_TRY(launch_dgemm(dn, DEPI_RESID, d->bf16w, st));
    }
    float* t = hin; hin = hout; hout = t;
  }
  *hfinal = hin;
  return ETD_OK;
}

// final LayerNorm + lm_head for `n` rows of X (fp32 [n][H]) -> logits [n][V]
int head_logits(etd_dec* d, const float* X, int n, float* logits, hipStream_t st) {
  if (!d->bf16w && n >= G3_MIN_ROWS && d->X1f && d->head.Wp && g3_enabled()) {
    ETD_TRY(launch_ln_rows_f32(X, n, d->H, d->lnfg, d->lnfb, nullptr, nullptr, d->cfg.layer_norm_eps, d->X1f, nullptr, st));
    DGemmArgs lm = g3_args(d->X1f, d->H, d->head, d->xf_log2, n);
    lm.bias = nullptr; lm.Y = logits; lm.ldy = d->V;
    return launch_gemm3(lm, DEPI_LOGITS, st);
  }
  DGemmArgs lm = {};
  lm.X = X; lm.ldx = d->H; lm.W = d->head.W; lm.bias = nullptr; lm.M = n; lm.N = d->V; lm.Npad = d->head.Npad; lm.K = d->H;
  lm.ln_g = d->lnfg; lm.ln_b = d->lnfb; lm.ln_eps = d->cfg.layer_norm_eps; lm.Y = logits; lm.ldy = d->V;
  lm.Wp = d->head.Wp; lm.w_log2 = d->head.w_log2; lm.x_log2 = d->xf_log2;
  return launch_dgemm(lm, DEPI_LOGITS, d->bf16w, st);
}

int check_slot(etd_dec* d, int slot) {
  if (!d || slot < 0 || slot >= d->S) ETD_FAIL(ETD_EINVAL, "decoder: bad slot %d", slot);
  return ETD_OK;
}

__global__ void k_init_slots(const int* __restrict__ init, int n, int* tgt_attrs, int* cur_tok, int* len, int* done, int* n_out, int* eos, int* limit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int* p = init + i * 7;
  const int s = p[0];
  tgt_attrs[4 * s] = p[1]; tgt_attrs[4 * s + 1] = p[2]; tgt_attrs[4 * s + 2] = p[3]; tgt_attrs[4 * s + 3] = p[4];
  cur_tok[s] = 0; len[s] = 0; done[s] = 0; n_out[s] = 0; eos[s] = p[5]; limit[s] = p[6];
}

// The attribute projection of a slot's TARGET attributes, ((t0 + t1) + t2) + t3 exactly as k_dembed / k_dstep_head add it
// (etude_decoder.py:171-176): fixed for the whole bar, so it is formed once here and every decode step reads one row
// instead of four.
__global__ __launch_bounds__(128) void k_slot_proj(const int* __restrict__ init, const float* __restrict__ attr_tab, int n_bins, int H, float* proj) {
  const int* p = init + blockIdx.x * 7;
  const int s = p[0];
  for (int k = threadIdx.x * 4; k < H; k += 128 * 4) {
    f32x4 t[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) t[a] = *reinterpret_cast<const f32x4*>(attr_tab + (long long)(a * n_bins + p[1 + a]) * H + k);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = ((t[0][c] + t[1][c]) + t[2][c]) + t[3][c];
    *reinterpret_cast<f32x4*>(proj + (long long)s * H + k) = o;
  }
}

// Stage a batch of n prompts (concatenated) on the device, embed them, run the model.  On return *hfinal holds
// the hidden states of all Mtot rows; the staged row metadata lives in d->ids (layout below).
struct Staged { int Mtot; const int *ids, *cls, *attrs, *row_slot, *row_pos, *row_active, *last_idx, *last_slot, *last_pos, *last_active, *init, *row_seq, *seq_row0, *seq_len; };

int stage_and_forward(etd_dec* d, int n, const int32_t* slots, const int32_t* T, const int32_t* ids, const int32_t* cls,
                      const int32_t* attrs4, const int32_t* init7 /* [n][7] or null */, Staged* sg, float** hfinal, hipStream_t st, bool* last_only = nullptr) {
  if (!d || n < 1 || n > d->S || !slots || !T || !ids || !cls || !attrs4) ETD_FAIL(ETD_EINVAL, "prefill: bad arguments");
  long long Mtot = 0;
  std::vector<char> seen((size_t)d->S, 0);
  for (int i = 0; i < n; ++i) {
    ETD_TRY(check_slot(d, slots[i]));
    if (T[i] <= 0 || T[i] > d->ctx) ETD_FAIL(ETD_EINVAL, "prefill: prompt %d has T=%d (max_ctx=%d)", i, T[i], d->ctx);
    if (seen[slots[i]]) ETD_FAIL(ETD_EINVAL, "prefill: slot %d listed twice", slots[i]);
    seen[slots[i]] = 1;
    Mtot += T[i];
  }
  if (Mtot > d->Mcap) ETD_FAIL(ETD_EINVAL, "prefill: %lld prompt rows exceed max_prefill_rows=%d", Mtot, d->Mcap);
  const int M = (int)Mtot;
  for (int i = 0; i < M; ++i) {
    if (ids[i] < 0 || ids[i] >= d->V || cls[i] < 0 || cls[i] >= d->cfg.num_classes) ETD_FAIL(ETD_EINVAL, "prefill: token/class id out of range at row %d", i);
    for (int k = 0; k < 4; ++k) if (attrs4[(size_t)k * M + i] < 0 || attrs4[(size_t)k * M + i] >= d->cfg.num_attribute_bins) ETD_FAIL(ETD_EINVAL, "prefill: attribute bin out of range at row %d", i);
  }
  // staging layout: [ids M][cls M][attrs 4M][row_slot M][row_pos M][row_active M][last_idx n][last_slot n][last_pos n][last_active n][init 7n]
  //                 [row_seq M][seq_row0 n][seq_len n]
  const size_t sv_ints = (size_t)10 * M + 13 * n;
  int* svp;
  const bool pinned = d->pin_stage && sv_ints <= d->pin_stage_ints;
  if (pinned) {
    HIP_TRY(hipEventSynchronize(d->pin_stage_evt));            // the previous upload (if any) has been read out of the buffer
    svp = d->pin_stage;                                         // (every word of the layout is written below: no memset of ~40 bytes per prompt row)
  } else {
    d->stage.resize(sv_ints);
    svp = d->stage.data();
  }
  memcpy(svp, ids, (size_t)M * 4);
  memcpy(svp + M, cls, (size_t)M * 4);
  memcpy(svp + 2 * (size_t)M, attrs4, (size_t)4 * M * 4);
  int* rs = svp + 6 * (size_t)M; int* rp = rs + M; int* ra = rp + M;
  int* li = ra + M; int* ls = li + n; int* lp = ls + n; int* la = lp + n; int* in7 = la + n;
  int* rq = in7 + 7 * n; int* sr0 = rq + M; int* sln = sr0 + n;
  int row = 0, max_len = 0; double kvb = 0, aflops = 0;
  for (int i = 0; i < n; ++i) {
    sr0[i] = row; sln[i] = T[i]; if (T[i] > max_len) max_len = T[i];
    aflops += 0.5 * 256.0 * d->nh * (double)T[i] * T[i];
    for (int t = 0; t < T[i]; ++t, ++row) { rs[row] = slots[i]; rp[row] = t; ra[row] = 1; rq[row] = i; }
    li[i] = row - 1; ls[i] = slots[i]; lp[i] = T[i] - 1; la[i] = 1;
    d->host_len[slots[i]] = T[i];
    kvb += (double)T[i] * d->nh * 64 * (2.0 * (d->bf16w ? 2 : 4) + 4 + 4);   // prefill: Q, K, V read once, O written once (K/V re-reads are L2 hits)
  }
  if (init7) memcpy(in7, init7, (size_t)7 * n * 4); else memset(in7, 0, (size_t)7 * n * 4);
  d->attn_bytes_hint = kvb;
  HIP_TRY(hipMemcpyAsync(d->ids, svp, sv_ints * 4, hipMemcpyHostToDevice, st));   // pinned: an async DMA; pageable fallback: returns after staging
  if (pinned) HIP_TRY(hipEventRecord(d->pin_stage_evt, st));
  d->host_n_out_valid = false;
  const int* b = d->ids;
  *sg = Staged{M, b, b + M, b + 2 * (size_t)M, b + 6 * (size_t)M, b + 7 * (size_t)M, b + 8 * (size_t)M, b + 9 * (size_t)M, b + 9 * (size_t)M + n,
               b + 9 * (size_t)M + 2 * n, b + 9 * (size_t)M + 3 * n, b + 9 * (size_t)M + 4 * n,
               b + 9 * (size_t)M + 11 * n, b + 10 * (size_t)M + 11 * n, b + 10 * (size_t)M + 12 * n};
  if (init7) {
    hipLaunchKernelGGL(k_init_slots, dim3((n + 63) / 64), dim3(64), 0, st, sg->init, n, d->tgt_attrs, d->cur_tok, d->len, d->done, d->n_out, d->eos, d->limit);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_slot_proj, dim3(n), dim3(128), 0, st, sg->init, d->attr_tab, d->cfg.num_attribute_bins, d->H, d->tgt_proj);
    HIP_TRY(hipGetLastError());
  }
  DEmbedArgs e = {};
  e.ids = sg->ids; e.cls = sg->cls; e.attrs = sg->attrs; e.M = M; e.H = d->H; e.n_bins = d->cfg.num_attribute_bins;
  e.word = d->word; e.cls_emb = d->cls_emb; e.attr_tab = d->attr_tab; e.h = d->h;
  e.rows = DecRows{sg->row_slot, sg->row_pos, sg->row_active, sg->row_seq};
  ETD_TRY(launch_dembed(e, st));
  PrefillInfo pf{n, sg->seq_row0, sg->seq_len, max_len, aflops};
  const bool can_mfma_attn = (d->bf16w ? d->Qb != nullptr : d->X1f != nullptr) && !ETD_XENV("ETD_NO_MFMA_PREFILL_ATTN");
  LastOnly lo{n, sg->last_idx, DecRows{sg->last_slot, sg->last_pos, sg->last_active}};
  ETD_TRY(forward_body(d, M, e.rows, hfinal, st, can_mfma_attn ? &pf : nullptr, false, last_only ? &lo : nullptr, last_only));
  return ETD_OK;
}

// KV cache, activation workspaces and stream state of one handle (everything that is not a weight)
int alloc_workspaces(etd_dec* d) {
  const int H = d->H;
  int rc = 0;
  d->slot_stride = (long long)d->nh * d->ctx * 64;
  d->layer_stride = d->slot_stride * d->S;
  const size_t kv_elems = (size_t)d->layer_stride * d->L;
  if (d->bf16w) { uint16_t *k, *v; if ((rc = d->alloc(&k, kv_elems, true)) || (rc = d->alloc(&v, kv_elems, true))) return rc; d->Kc = k; d->Vc = v; }
  else { float *k, *v; if ((rc = d->alloc(&k, kv_elems, true)) || (rc = d->alloc(&v, kv_elems, true))) return rc; d->Kc = k; d->Vc = v; }
  const size_t M = d->Mcap;
  const size_t Mf = d->bf16w ? (size_t)(d->S > 1 ? d->S : 1) : M;     // rows that can take the fp32-activation path
  rc = 0;
  rc = rc ? rc : d->alloc(&d->h, M * H); rc = rc ? rc : d->alloc(&d->h2, M * H);
  rc = rc ? rc : d->alloc(&d->Q, M * H); rc = rc ? rc : d->alloc(&d->AO, M * H); rc = rc ? rc : d->alloc(&d->DO, M * H);
  rc = rc ? rc : d->alloc(&d->M1, Mf * d->I); rc = rc ? rc : d->alloc(&d->logits, (size_t)d->Mmax * d->V);
  rc = rc ? rc : d->alloc(&d->qkv_raw, (size_t)3 * H); rc = rc ? rc : d->alloc(&d->hlast, (size_t)d->S * H);
  if (d->bf16w) {
    rc = rc ? rc : d->alloc(&d->X1b, M * H); rc = rc ? rc : d->alloc(&d->X2b, M * H); rc = rc ? rc : d->alloc(&d->AOb, M * H);
    rc = rc ? rc : d->alloc(&d->M1b, M * d->I);
    rc = rc ? rc : d->alloc(&d->Pk, (size_t)12 * DS_STEP_MAX_ROWS * H);     // split-K slabs of the decode step: 5 (down | dense) or 4 (down) + one per head (dense inside the attention workgroups)
    rc = rc ? rc : d->alloc(&d->row_cnt, (size_t)d->L * DS_STEP_MAX_ROWS, true);
    rc = rc ? rc : d->alloc(&d->Xcat, M * (d->I + H));
    rc = rc ? rc : d->alloc(&d->Qb, M * H);      // RoPE'd queries of a batched prefill (K / V: the cache rows)
  }
  if (!d->bf16w) {
    rc = rc ? rc : d->alloc(&d->X1f, M * H); rc = rc ? rc : d->alloc(&d->X2f, M * H);
    const size_t Sr = d->S > 1 ? d->S : 1;
    rc = rc ? rc : d->alloc(&d->t_q, Sr * H); rc = rc ? rc : d->alloc(&d->t_ao, Sr * H); rc = rc ? rc : d->alloc(&d->t_do, Sr * H); rc = rc ? rc : d->alloc(&d->t_m1, Sr * d->I);
  }
  rc = rc ? rc : d->alloc(&d->samp_dev, (size_t)1, true); rc = rc ? rc : d->alloc(&d->rng_key, (size_t)d->S, true);
  rc = rc ? rc : d->alloc(&d->row_sp, (size_t)2 * d->Mmax, true);
  rc = rc ? rc : d->alloc(&d->row_slot, (size_t)d->Mmax); rc = rc ? rc : d->alloc(&d->row_pos, (size_t)d->Mmax); rc = rc ? rc : d->alloc(&d->row_active, (size_t)d->Mmax);
  rc = rc ? rc : d->alloc(&d->ids, 10 * M + 13 * (size_t)d->S); rc = rc ? rc : d->alloc(&d->slots_dev, (size_t)d->S);
  const size_t S = d->S;
  rc = rc ? rc : d->alloc(&d->cur_tok, S, true); rc = rc ? rc : d->alloc(&d->len, S, true); rc = rc ? rc : d->alloc(&d->done, S, true);
  rc = rc ? rc : d->alloc(&d->n_out, S, true); rc = rc ? rc : d->alloc(&d->eos, S, true); rc = rc ? rc : d->alloc(&d->limit, S, true);
  rc = rc ? rc : d->alloc(&d->tgt_proj, S * (size_t)d->H, true);
  rc = rc ? rc : d->alloc(&d->tgt_attrs, 4 * S, true); rc = rc ? rc : d->alloc(&d->out_tok, S * d->out_cap, true);
  if (!rc && !ETD_XENV("ETD_NO_PINNED")) {
    d->pin_stage_ints = 10 * M + 13 * S;
    if (hipHostMalloc((void**)&d->pin_stage, d->pin_stage_ints * 4, hipHostMallocDefault) != hipSuccess) { d->pin_stage = nullptr; (void)hipGetLastError(); }
    if (hipHostMalloc((void**)&d->pin_rb, (2 * S + S * (size_t)d->out_cap) * 4, hipHostMallocDefault) != hipSuccess) { d->pin_rb = nullptr; (void)hipGetLastError(); }
    if (d->pin_stage && hipEventCreateWithFlags(&d->pin_stage_evt, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(d->pin_stage); d->pin_stage = nullptr; d->pin_stage_evt = nullptr; }
  }
  d->host_n_out.assign(S, 0);
  return rc;
}

}  // namespace

// 1 when the library was built with -DETD_EXPERIMENTS: the measured dead ends (in-launch row finish, 8 / 16-wave attention workgroups, k_post_attn) are compiled in
// and their environment switches are live; the shipped build has one path per precision and ignores those switches
extern "C" int etd_has_experiments(void) {
#ifdef ETD_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

extern "C" int etd_decoder_operand_type(void) { return ETD_DEC_IS_F16; }

extern "C" int etd_decoder_create(const etd_dec_cfg* cfg, const char* const* names, const float* const* host_ptrs,
                                  const int64_t* numels, int n, etd_dec** out) {
  if (!cfg || !names || !host_ptrs || !numels || !out) ETD_FAIL(ETD_EINVAL, "decoder_create: null argument");
  if (cfg->struct_bytes != (int)sizeof(etd_dec_cfg)) ETD_FAIL(ETD_EINVAL, "decoder_create: etd_dec_cfg of %d bytes, this library (ABI %d) expects %d -- caller built against another etude_hip.h", cfg->struct_bytes, ETD_ABI_VERSION, (int)sizeof(etd_dec_cfg));
  const etd_dec_cfg& c = *cfg;
  if (c.hidden_size % 256 || c.num_attention_heads <= 0 || c.hidden_size / c.num_attention_heads != 64 || c.intermediate_size % 128 ||
      (int)(64 * c.rotary_pct) != 16 || c.num_hidden_layers < 1 || c.vocab_size < 2 || c.max_streams < 1 || c.max_ctx < 16 ||
      c.attribute_emb_dim < 1 || c.num_attribute_bins < 1)
    ETD_FAIL(ETD_EINVAL, "decoder_create: unsupported config (need head_dim 64, rotary_ndims 16, hidden %% 256 == 0, intermediate %% 128 == 0)");
  etd_dec* d = new etd_dec();
  d->cfg = c; d->bf16w = c.precision == 1;
  d->H = c.hidden_size; d->I = c.intermediate_size; d->V = c.vocab_size; d->L = c.num_hidden_layers; d->nh = c.num_attention_heads;
  d->S = c.max_streams; d->ctx = c.max_ctx; d->out_cap = 1024;
  d->Mmax = d->ctx > d->S ? d->ctx : d->S;
  d->Mcap = c.max_prefill_rows > d->Mmax ? c.max_prefill_rows : d->Mmax;
  d->host_len.assign(d->S, 0);
  d->host_key.resize(d->S);
  for (int i = 0; i < d->S; ++i) d->host_key[i] = (unsigned long long)i;
  Loader Ld;
  for (int i = 0; i < n; ++i) Ld.t[names[i]] = {host_ptrs[i], numels[i]};
  auto fail = [&](int rc) {
    for (void* p : d->allocs) (void)hipFree(p);
    if (d->pin_stage) (void)hipHostFree(d->pin_stage);
    if (d->pin_rb) (void)hipHostFree(d->pin_rb);
    if (d->pin_stage_evt) (void)hipEventDestroy(d->pin_stage_evt);
    delete d; return rc;
  };
  const int H = d->H, E = c.attribute_emb_dim, NB = c.num_attribute_bins;
  int rc;
  if ((rc = load_vec(d, Ld, "word_embeddings.weight", d->V * H, &d->word))) return fail(rc);
  if ((rc = load_vec(d, Ld, "class_embeddings.weight", c.num_classes * H, &d->cls_emb))) return fail(rc);
  {
    // attribute_projection(cat(e0,e1,e2,e3)) = bias + sum_a W[:, aE:(a+1)E] e_a  -> per (attribute, bin) vectors
    const char* an[4] = {"pitch_overlap_embeddings.weight", "polyphony_embeddings.weight", "note_sustain_embeddings.weight", "rhythm_intensity_embeddings.weight"};
    const float* pw = Ld.get("attribute_projection.weight", (int64_t)H * 4 * E);
    const float* pb = Ld.get("attribute_projection.bias", H);
    if (!pw || !pb) return fail(ETD_EINVAL);
    std::vector<float> tab((size_t)4 * NB * H);
    for (int a = 0; a < 4; ++a) {
      const float* em = Ld.get(an[a], (int64_t)NB * E);
      if (!em) return fail(ETD_EINVAL);
      for (int b = 0; b < NB; ++b)
        for (int o = 0; o < H; ++o) {
          double s = (a == 0) ? (double)pb[o] : 0.0;
          for (int e = 0; e < E; ++e) s += (double)pw[(size_t)o * 4 * E + a * E + e] * em[b * E + e];
          tab[((size_t)a * NB + b) * H + o] = (float)s;
        }
    }
    if ((rc = up_f32(d, &d->attr_tab, tab.data(), tab.size()))) return fail(rc);
  }
  d->layers.resize(d->L);
  for (int l = 0; l < d->L; ++l) {
    const std::string p = "transformer.layers." + std::to_string(l) + ".";
    Layer& w = d->layers[l];
    if ((rc = load_vec(d, Ld, p + "input_layernorm.weight", H, &w.ln1g))) return fail(rc);
    if ((rc = load_vec(d, Ld, p + "input_layernorm.bias", H, &w.ln1b))) return fail(rc);
    if ((rc = load_vec(d, Ld, p + "post_attention_layernorm.weight", H, &w.ln2g))) return fail(rc);
    if ((rc = load_vec(d, Ld, p + "post_attention_layernorm.bias", H, &w.ln2b))) return fail(rc);
    if ((rc = load_lin(d, Ld, p + "attention.query_key_value", 3 * H, H, true, &w.qkv))) return fail(rc);
    if ((rc = load_lin(d, Ld, p + "attention.dense", H, H, true, &w.dense))) return fail(rc);
    if ((rc = load_lin(d, Ld, p + "mlp.dense_h_to_4h", d->I, H, true, &w.up))) return fail(rc);
    if ((rc = load_lin(d, Ld, p + "mlp.dense_4h_to_h", H, d->I, true, &w.down))) return fail(rc);
    if (!d->bf16w) {
      // plane scales of the fp32-grade f16 path from provable bounds (csrc/gemm3.h): LayerNorm outputs by their parameters, projections of them by Cauchy-Schwarz on the
      // weight rows; the rotary embedding mixes two dims of a row (|x1 c - x2 s| <= |x1| + |x2|); the attention output is a convex combination of V rows; |gelu(u)| <= |u|
      const float* g1 = Ld.get(p + "input_layernorm.weight", H); const float* b1 = Ld.get(p + "input_layernorm.bias", H);
      const float* g2 = Ld.get(p + "post_attention_layernorm.weight", H); const float* b2 = Ld.get(p + "post_attention_layernorm.bias", H);
      const float* Wq = Ld.get(p + "attention.query_key_value.weight", (int64_t)3 * H * H); const float* bq = Ld.get(p + "attention.query_key_value.bias", 3 * H);
      const float* Wu = Ld.get(p + "mlp.dense_h_to_4h.weight", (int64_t)d->I * H); const float* bu = Ld.get(p + "mlp.dense_h_to_4h.bias", d->I);
      if (!g1 || !b1 || !g2 || !b2 || !Wq || !bq || !Wu || !bu) return fail(ETD_EINVAL);
      w.x1_log2 = g3_scale_log2(g3_bound_ln(g1, b1, H));
      w.x2_log2 = g3_scale_log2(g3_bound_ln(g2, b2, H));
      std::vector<float> rb((size_t)3 * H);
      g3_row_bounds_of_ln(Wq, bq, 3 * H, H, g1, b1, rb.data());
      float bnd[3] = {0.f, 0.f, 0.f};
      for (int j = 0; j < 3 * H; ++j) { const int part = (j % 192) >> 6; bnd[part] = fmaxf(bnd[part], rb[j]); }
      w.q_log2 = g3_scale_log2(2.f * bnd[0]); w.k_log2 = g3_scale_log2(2.f * bnd[1]); w.v_log2 = g3_scale_log2(bnd[2]);
      w.m_log2 = g3_scale_log2(g3_bound_linear_of_ln(Wu, bu, d->I, H, g2, b2));
    }
#if ETD_DEC_IS_F16
    if (d->bf16w) {
      // The 16-bit mode's operands are IEEE half (dec_kernels.h): unlike bf16 they END at 65 504, and an Inf in a LayerNorm row, a cache row or the hidden layer becomes NaN in
      // the softmax / the next LayerNorm.  Every 16-bit tensor of this path is a weight, a LayerNorm output, a projection of one (Q / K / V after the rotary mix, GELU(up): |gelu(u)| <= |u|)
      // or a convex combination of V rows: the same provable bounds the fp32-grade path takes its plane scales from (csrc/gemm3.h) say at load time whether THIS checkpoint
      // can leave the range, whatever the input.  A checkpoint that could is refused (a -DETD_DEC_BF16 build, or precision "fp32", takes it).
      const float* g1 = Ld.get(p + "input_layernorm.weight", H); const float* b1 = Ld.get(p + "input_layernorm.bias", H);
      const float* g2 = Ld.get(p + "post_attention_layernorm.weight", H); const float* b2l = Ld.get(p + "post_attention_layernorm.bias", H);
      const float* Wq = Ld.get(p + "attention.query_key_value.weight", (int64_t)3 * H * H); const float* bq = Ld.get(p + "attention.query_key_value.bias", 3 * H);
      const float* Wu = Ld.get(p + "mlp.dense_h_to_4h.weight", (int64_t)d->I * H); const float* bu = Ld.get(p + "mlp.dense_h_to_4h.bias", d->I);
      if (!g1 || !b1 || !g2 || !b2l || !Wq || !bq || !Wu || !bu) return fail(ETD_EINVAL);
      std::vector<float> rb((size_t)3 * H);
      g3_row_bounds_of_ln(Wq, bq, 3 * H, H, g1, b1, rb.data());
      float qkv_b = 0.f;
      for (int j = 0; j < 3 * H; ++j) qkv_b = fmaxf(qkv_b, (((j % 192) >> 6) < 2 ? 2.f : 1.f) * rb[j]);      // (the rotary embedding mixes two dims of a Q / K row)
      const float bounds[4] = {g3_bound_ln(g1, b1, H), g3_bound_ln(g2, b2l, H), qkv_b, g3_bound_linear_of_ln(Wu, bu, d->I, H, g2, b2l)};
      static const char* const what[4] = {"input_layernorm rows", "post_attention_layernorm rows", "Q / K / V rows", "gelu(dense_h_to_4h) rows"};
      for (int i = 0; i < 4; ++i)
        if (!(bounds[i] < 65504.f)) { g_etd_err = "decoder_create: layer " + std::to_string(l) + ": " + what[i] + " can reach " + std::to_string(bounds[i]) +
                                                  " (provable bound), beyond the IEEE-half range of the 16-bit serving mode; use precision \"fp32\" or a -DETD_DEC_BF16 build"; return fail(ETD_EINVAL); }
    }
#endif
    if (d->bf16w) {
      // h_new - h = W2 gelu(..) + b2 + Wd attn + bd  ==  [W2 | Wd] [gelu(..) ; attn] + (b2 + bd)
      const float* W2 = Ld.get(p + "mlp.dense_4h_to_h.weight", (int64_t)H * d->I);
      const float* Wd = Ld.get(p + "attention.dense.weight", (int64_t)H * H);
      const float* b2 = Ld.get(p + "mlp.dense_4h_to_h.bias", H);
      const float* bd = Ld.get(p + "attention.dense.bias", H);
      const int Kc = d->I + H;
      std::vector<uint16_t> wc((size_t)H * Kc);
      for (int o = 0; o < H; ++o) {
        for (int k = 0; k < d->I; ++k) wc[(size_t)o * Kc + k] = f2bf_h(W2[(size_t)o * d->I + k]);
        for (int k = 0; k < H; ++k) wc[(size_t)o * Kc + d->I + k] = f2bf_h(Wd[(size_t)o * H + k]);
      }
      uint16_t* pw; if ((rc = d->alloc(&pw, wc.size()))) return fail(rc);
      HIP_TRY(hipMemcpy(pw, wc.data(), wc.size() * 2, hipMemcpyHostToDevice));
      std::vector<float> bc(H);
      for (int o = 0; o < H; ++o) bc[o] = b2[o] + bd[o];
      w.cat.W = pw; w.cat.N = H; w.cat.Npad = H; w.cat.K = Kc;
      {
        std::vector<uint16_t> dh((size_t)d->nh * H * 64);
        for (int hd = 0; hd < d->nh; ++hd)
          for (int o = 0; o < H; ++o)
            for (int dd = 0; dd < 64; ++dd) dh[((size_t)hd * H + o) * 64 + dd] = f2bf_h(Wd[(size_t)o * H + hd * 64 + dd]);
        uint16_t* pd; if ((rc = d->alloc(&pd, dh.size()))) return fail(rc);
        HIP_TRY(hipMemcpy(pd, dh.data(), dh.size() * 2, hipMemcpyHostToDevice));
        w.dense_hw = pd;
      }
      if (H % 32 == 0 && Kc % 16 == 0) {
        std::vector<uint16_t> wp(wc.size());
        pack_wfrag_host(wc.data(), H, Kc, wp.data());
        uint16_t* pf; if ((rc = d->alloc(&pf, wp.size()))) return fail(rc);
        HIP_TRY(hipMemcpy(pf, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
        w.cat.Wf = pf;
      }
      if ((rc = up_f32(d, &w.cat.b, bc.data(), H))) return fail(rc);
      if (H == 512 && d->I == 2048 && fused_pmlp_on()) {
        const float* W1 = Ld.get(p + "mlp.dense_h_to_4h.weight", (int64_t)d->I * H);
        if (!W1) return fail(ETD_EINVAL);
        std::vector<uint16_t> w1((size_t)d->I * H), ws((size_t)DMLP_STREAM_ELEMS);
        for (size_t i = 0; i < w1.size(); ++i) w1[i] = f2bf_h(W1[i]);
        pack_dmlp_weights(w1.data(), wc.data(), ws.data());
        uint16_t* pm; if ((rc = d->alloc(&pm, ws.size()))) return fail(rc);
        HIP_TRY(hipMemcpy(pm, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
        w.mlp_frag = pm;
      }
    }
  }
  if ((rc = load_vec(d, Ld, "transformer.final_layer_norm.weight", H, &d->lnfg))) return fail(rc);
  if ((rc = load_vec(d, Ld, "transformer.final_layer_norm.bias", H, &d->lnfb))) return fail(rc);
  if ((rc = load_lin(d, Ld, "lm_head", d->V, H, false, &d->head))) return fail(rc);
  if (!d->bf16w) {
    const float* gf = Ld.get("transformer.final_layer_norm.weight", H); const float* bf_ = Ld.get("transformer.final_layer_norm.bias", H);
    if (!gf || !bf_) return fail(ETD_EINVAL);
    d->xf_log2 = g3_scale_log2(g3_bound_ln(gf, bf_, H));
  }
  if (d->bf16w && H % 16 == 0) {
    // the same d16 values in the order one wave's A-operand loads want them: (tile t, k-step s, lane l) holds row
    // 32 t + (l & 31), columns 16 s + 8 (l >> 5) .. +8 -- each load instruction then reads one contiguous 1 KiB block
    // instead of 32 B out of 32 different rows (the decode-step head kernel lives on one CU per 32 streams: its L1 traffic counts)
    const float* W = Ld.get("lm_head.weight", (int64_t)d->V * H);
    if (!W) return fail(ETD_EINVAL);
    const int tiles = d->head.Npad / 32, ks = H / 16;
    std::vector<uint16_t> hp((size_t)tiles * ks * 64 * 8, 0);
    for (int t = 0; t < tiles; ++t)
      for (int sx = 0; sx < ks; ++sx)
        for (int l = 0; l < 64; ++l) {
          const int row = t * 32 + (l & 31);
          if (row >= d->V) continue;
          for (int e = 0; e < 8; ++e) hp[(((size_t)t * ks + sx) * 64 + l) * 8 + e] = f2bf_h(W[(size_t)row * H + sx * 16 + (l >> 5) * 8 + e]);
        }
    uint16_t* pp; if ((rc = d->alloc(&pp, hp.size()))) return fail(rc);
    if (hipMemcpy(pp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return fail(ETD_EHIP);
    d->head_frag = pp;
  }
  {
    // RoPE tables as HF builds them in fp32: inv_freq = 1/theta^(2i/rot), angle = pos * inv_freq (modeling_gpt_neox.py:72-107)
    std::vector<float> cs((size_t)d->ctx * 8), sn((size_t)d->ctx * 8);
    for (int i = 0; i < 8; ++i) {
      const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / 16.0f);
      for (int p = 0; p < d->ctx; ++p) { const float ang = (float)p * inv; cs[(size_t)p * 8 + i] = cosf(ang); sn[(size_t)p * 8 + i] = sinf(ang); }
    }
    if ((rc = up_f32(d, &d->rope_cos, cs.data(), cs.size()))) return fail(rc);
    if ((rc = up_f32(d, &d->rope_sin, sn.data(), sn.size()))) return fail(rc);
  }
  d->n_weight_allocs = d->allocs.size();
  rc = alloc_workspaces(d);
  if (rc) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { g_etd_err = "decoder_create: device synchronisation failed"; return fail(ETD_EHIP); }
  *out = d;
  return ETD_OK;
}

// clone / destroy bookkeeping of a weight-sharing family (n_clones, zombie) runs under one process-wide lock: handles of a family
// may be cloned and destroyed from different threads
static std::mutex g_family_mu;

extern "C" int etd_decoder_clone(etd_dec* src, etd_dec** out) {
  if (!src || !out) ETD_FAIL(ETD_EINVAL, "decoder_clone: null argument");
  etd_dec* own = src->weights_owner ? src->weights_owner : src;
  {
    std::lock_guard<std::mutex> lk(g_family_mu);
    if (own->zombie) ETD_FAIL(ETD_EINVAL, "decoder_clone: the source handle was destroyed");
    ++own->n_clones;          // taken BEFORE the workspaces are built: the owner cannot free the weights underneath this clone
  }
  auto unref = [own]() {
    bool last;
    { std::lock_guard<std::mutex> lk(g_family_mu); last = --own->n_clones == 0 && own->zombie; }
    if (last) { for (void* p : own->allocs) (void)hipFree(p); delete own; }
  };
  etd_dec* d = new etd_dec();
  d->cfg = own->cfg; d->bf16w = own->bf16w;
  d->H = own->H; d->I = own->I; d->V = own->V; d->L = own->L; d->nh = own->nh; d->S = own->S; d->ctx = own->ctx;
  d->Mmax = own->Mmax; d->Mcap = own->Mcap; d->out_cap = own->out_cap;
  d->word = own->word; d->cls_emb = own->cls_emb; d->attr_tab = own->attr_tab; d->layers = own->layers;
  d->lnfg = own->lnfg; d->lnfb = own->lnfb; d->head = own->head; d->head_frag = own->head_frag; d->xf_log2 = own->xf_log2;
  d->rope_cos = own->rope_cos; d->rope_sin = own->rope_sin;
  d->host_len.assign(d->S, 0);
  d->host_key.resize(d->S);
  for (int i = 0; i < d->S; ++i) d->host_key[i] = (unsigned long long)i;
  d->weights_owner = own;
  int rc = alloc_workspaces(d);
  if (!rc && hipDeviceSynchronize() != hipSuccess) { g_etd_err = "decoder_clone: device synchronisation failed"; rc = ETD_EHIP; }
  if (rc) {
    for (void* p : d->allocs) (void)hipFree(p);
    if (d->pin_stage) (void)hipHostFree(d->pin_stage);
    if (d->pin_rb) (void)hipHostFree(d->pin_rb);
    if (d->pin_stage_evt) (void)hipEventDestroy(d->pin_stage_evt);
    delete d; unref(); return rc;
  }
  *out = d;
  return ETD_OK;
}

extern "C" void etd_decoder_destroy(etd_dec* d) {
  if (!d) return;
  (void)hipDeviceSynchronize();   // kernels of this handle may still be in flight
  for (auto& kv : d->graphs) (void)hipGraphExecDestroy(kv.second);
  d->graphs.clear();
  if (d->trace) { (void)hipFree(d->trace); d->trace = nullptr; }
  if (d->trace_step) { (void)hipFree(d->trace_step); d->trace_step = nullptr; }
  if (d->trace_pk) { (void)hipFree(d->trace_pk); d->trace_pk = nullptr; }
  if (d->trace_q) { (void)hipFree(d->trace_q); d->trace_q = nullptr; }
  if (d->trace_dbg) { (void)hipFree(d->trace_dbg); d->trace_dbg = nullptr; }
  if (d->stamp_dev) { (void)hipFree(d->stamp_dev); d->stamp_dev = nullptr; }
  if (d->logits_dbg) { (void)hipFree(d->logits_dbg); d->logits_dbg = nullptr; }
  if (d->pin_stage) { (void)hipHostFree(d->pin_stage); d->pin_stage = nullptr; }
  if (d->pin_rb) { (void)hipHostFree(d->pin_rb); d->pin_rb = nullptr; }
  if (d->pin_stage_evt) { (void)hipEventDestroy(d->pin_stage_evt); d->pin_stage_evt = nullptr; }
  if (d->weights_owner) {
    etd_dec* own = d->weights_owner;
    for (void* p : d->allocs) (void)hipFree(p);
    delete d;
    bool last;
    { std::lock_guard<std::mutex> lk(g_family_mu); last = --own->n_clones == 0 && own->zombie; }
    if (last) { for (void* p : own->allocs) (void)hipFree(p); delete own; }
    return;
  }
  bool keep;
  { std::lock_guard<std::mutex> lk(g_family_mu); keep = d->n_clones > 0; if (keep) d->zombie = true; }
  if (keep) {
    // clones still read the weights: release this handle's own workspaces now, the weights with the last clone
    for (size_t i = d->n_weight_allocs; i < d->allocs.size(); ++i) (void)hipFree(d->allocs[i]);
    d->allocs.resize(d->n_weight_allocs);
    return;
  }
  for (void* p : d->allocs) (void)hipFree(p);
  delete d;
}

extern "C" int etd_decoder_begin_bars(etd_dec* d, int n, const int32_t* slots, const int32_t* T, const int32_t* ids, const int32_t* cls,
                                      const int32_t* attrs4, const int32_t* tgt_attrs4, const int32_t* eos_ids, const int32_t* limits, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || n < 1 || !T || !tgt_attrs4 || !eos_ids || !limits) ETD_FAIL(ETD_EINVAL, "begin_bars: bad arguments");
  std::vector<int> init((size_t)7 * n);
  for (int i = 0; i < n; ++i) {
    if (limits[i] < 1 || limits[i] > d->out_cap) ETD_FAIL(ETD_EINVAL, "begin_bars: limit %d outside [1, %d]", limits[i], d->out_cap);
    // positions a bar touches: the prompt's T, then one per generated token that is fed back (all but the last) -- past max_ctx the
    // K/V rows could not be stored and the RoPE table would be read out of bounds
    if ((long long)T[i] + limits[i] - 1 > d->ctx) ETD_FAIL(ETD_EINVAL, "begin_bars: prompt of %d tokens + limit %d needs %d KV positions, max_ctx is %d", T[i], limits[i], T[i] + limits[i] - 1, d->ctx);
    for (int k = 0; k < 4; ++k) if (tgt_attrs4[4 * i + k] < 0 || tgt_attrs4[4 * i + k] >= d->cfg.num_attribute_bins) ETD_FAIL(ETD_EINVAL, "begin_bars: target attribute out of range");
    init[7 * i] = slots ? slots[i] : 0;
    for (int k = 0; k < 4; ++k) init[7 * i + 1 + k] = tgt_attrs4[4 * i + k];
    init[7 * i + 5] = eos_ids[i]; init[7 * i + 6] = limits[i];
  }
  if (d->sampling && d->keys_dirty) {
    HIP_TRY(hipMemcpyAsync(d->rng_key, d->host_key.data(), (size_t)d->S * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    d->keys_dirty = false;
  }
  // the row-finish counters (ETD_ROWFIN=1) are zero between launches by construction (the last arriver resets its word); a launch
  // that died half way would leave them poisoned for good, so every bar starts from zero anyway (16 KiB, on the stream)
#ifdef ETD_EXPERIMENTS
  static const bool rowfin_on = getenv("ETD_ROWFIN") && atoi(getenv("ETD_ROWFIN")) > 0;
#else
  constexpr bool rowfin_on = false;
#endif
  if (rowfin_on && d->row_cnt) HIP_TRY(hipMemsetAsync(d->row_cnt, 0, (size_t)d->L * DS_STEP_MAX_ROWS * sizeof(int), st));
  Staged sg; float* hf = nullptr;
  bool compact = false;
  ETD_TRY(stage_and_forward(d, n, slots, T, ids, cls, attrs4, init.data(), &sg, &hf, st, &compact));
  // only each prompt's last position feeds the first generated token (etude_decoder.py:317)
  if (compact) {
    ETD_TRY(head_logits(d, hf, n, d->logits, st));           // (the last layer already ran on those rows alone)
  } else {
    ETD_TRY(launch_gather_rows(hf, sg.last_idx, n, d->H, d->hlast, st));
    ETD_TRY(head_logits(d, d->hlast, n, d->logits, st));
  }
  DArgmaxArgs am = {};
  am.logits = d->logits; am.ldl = d->V; am.V = d->V; am.M = n;
  am.rows = DecRows{sg.last_slot, sg.last_pos, sg.last_active};
  am.cur_tok = d->cur_tok; am.len = d->len; am.done = d->done; am.n_out = d->n_out; am.out_tok = d->out_tok; am.out_cap = d->out_cap;
  am.eos = d->eos; am.limit = d->limit;
  if (d->sampling) { am.samp = d->samp_dev; am.rng_key = d->rng_key; }
  ETD_TRY(launch_dargmax(am, st));
  return ETD_OK;
}

extern "C" int etd_decoder_set_sampling(etd_dec* d, float temperature, float top_p, unsigned long long seed, void* stream) {
  if (!d || !(temperature >= 0.f) || !(top_p == top_p)) ETD_FAIL(ETD_EINVAL, "set_sampling: bad arguments");
  if (temperature > 0.f && d->V > 256) ETD_FAIL(ETD_EINVAL, "set_sampling: sampling supports vocabularies up to 256 entries");
  DSampleCfg c; c.inv_temp = temperature > 0.f ? 1.0f / temperature : 0.f; c.top_p = top_p; c.seed = seed;
  HIP_TRY(hipMemcpyAsync(d->samp_dev, &c, sizeof(c), hipMemcpyHostToDevice, (hipStream_t)stream));   // pageable source: staged before return
  d->sampling = temperature > 0.f;
  return ETD_OK;
}

extern "C" int etd_decoder_set_keys(etd_dec* d, int n, const int32_t* slots, const unsigned long long* keys) {
  if (!d || n < 1 || !slots || !keys) ETD_FAIL(ETD_EINVAL, "set_keys: bad arguments");
  for (int i = 0; i < n; ++i) { ETD_TRY(check_slot(d, slots[i])); d->host_key[slots[i]] = keys[i]; }
  d->keys_dirty = true;
  return ETD_OK;
}

extern "C" int etd_decoder_begin_bar(etd_dec* d, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                                     const int32_t* tgt_attrs4, int eos_id, int limit, void* stream) {
  const int32_t sl = slot, tt = T, eo = eos_id, li = limit;
  return etd_decoder_begin_bars(d, 1, &sl, &tt, ids, cls, attrs4, tgt_attrs4, &eo, &li, stream);
}

extern "C" int etd_decoder_step(etd_dec* d, const int32_t* slots, int n_active, int n_steps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !slots || n_active < 1 || n_active > d->S || n_steps < 1) ETD_FAIL(ETD_EINVAL, "decoder_step: bad args");
  for (int i = 0; i < n_active; ++i) ETD_TRY(check_slot(d, slots[i]));
  if ((int)d->last_slots.size() != n_active || memcmp(d->last_slots.data(), slots, (size_t)n_active * 4)) {
    d->last_slots.assign(slots, slots + n_active);
    HIP_TRY(hipMemcpyAsync(d->slots_dev, d->last_slots.data(), (size_t)n_active * 4, hipMemcpyHostToDevice, st));
  }
  d->host_n_out_valid = false;
  {
    // Two rows of a head per attention workgroup (dec_kernels.hip, launch_dstep_attn_down) or one: BIT-IDENTICAL results
    // (tests/test_gpu_decoder.py::test_paired_rows_form_is_bit_identical), so the choice is a pure performance matter -- and it is a
    // deterministic function of (rows, mean context) alone, part of the captured graph's key.  Measured on MI355X (round 2: tools/README.md "pair
    // sweeps"; round 3 at 432 rows: profiles/r03_*): the pair form pays from 32 rows up at mean contexts 192 .. 1024 unless the launch's
    // workgroups quantise badly over the 256 CUs (64 + 8 M one-row workgroups of 1 unit against 32 + 4 M pair workgroups of 2 units: the
    // busiest CU decides), which matters up to ~128 rows only.
    double ctx_sum = 0;
    for (int i = 0; i < n_active; ++i) ctx_sum += d->host_len[slots[i]] + 1;
    const double mean_ctx = ctx_sum / n_active;
    bool pair = n_active >= 32 && mean_ctx >= 192.0 && mean_ctx <= 1024.0;
    if (pair && n_active <= 128) {
      const int u1 = (64 + 8 * n_active + 255) / 256, u2 = 2 * ((32 + 4 * n_active + 255) / 256);
      pair = u2 <= u1;
    }
    d->step_pair = d->force_pair < 0 ? pair : (d->force_pair > 0 && n_active >= 2);
  }
  d->rows_identity = true;
  for (int i = 0; i < n_active; ++i) if (slots[i] != i) { d->rows_identity = false; break; }
  // d16 batched decode step on the fused kernels: [embed + LayerNorm] once per call, then per step 4 launches per layer
  // (QKV|up, attention, down|dense, residual + LayerNorm) and one head launch that also prepares the next step's rows
  const int vpad = (d->V + 31) / 32 * 32;
  // A single stream steps on the fused kernels too (round 3; ETD_FUSED_M1=0: the GEMV sequence of rounds 1-2): 25 launches per step instead of ~36 and the same
  // arithmetic as inside a batch -- one job of 92 bars x 48 tokens 1.09 -> 0.66 s (tools/runs3/r3_run20.sh), every decoder golden unchanged.
  static const bool fused_m1 = !(ETD_XENV("ETD_FUSED_M1") && atoi(ETD_XENV("ETD_FUSED_M1")) == 0);
  const bool fused = d->bf16w && (n_active > 1 || fused_m1) && n_active <= DS_STEP_MAX_ROWS && (d->I + d->H) % (5 * 64 * 8) == 0 && d->H == 512 && vpad <= 256 &&
                     vpad <= d->head.Npad && d->head_frag && !ETD_XENV("ETD_NO_FUSED_STEP");
  d->last_step_fused = fused;
  d->stamp_on = d->stamp_armed && d->stamp_skip <= 0;          // (a whole call is stamped or not: the scheduler issues one bar's steps per call)
  if (d->stamp_armed && d->stamp_skip > 0) d->stamp_skip -= n_steps;
  auto embed = [&](hipStream_t s_) -> int {
    DEmbedArgs e = {};
    e.slots = d->slots_dev; e.len = d->len; e.done = d->done; e.row_slot_out = d->row_slot; e.row_pos_out = d->row_pos; e.row_active_out = d->row_active; e.row_sp_out = d->row_sp;
    e.cur_tok = d->cur_tok; e.tgt_attrs = d->tgt_attrs; e.tgt_cls = 2 /* TGT_CLASS_ID, etude/data/dataset.py:19 */;
    e.M = n_active; e.H = d->H; e.n_bins = d->cfg.num_attribute_bins;
    e.word = d->word; e.cls_emb = d->cls_emb; e.attr_tab = d->attr_tab; e.h = d->h;
    e.rows = DecRows{d->row_slot, d->row_pos, d->row_active};
    return launch_dembed(e, s_);
  };
  // K and V rows one layer's attention launch reads at step `s` of this call: every row's context so far (+ the row it appends)
  const double kv_row_bytes = (double)d->nh * 64 * 2 * (d->bf16w ? 2 : 4);
  auto step_kv_bytes = [&](int s) {
    double b = 0;
    for (int i = 0; i < n_active; ++i) { const int c = d->host_len[slots[i]] + 1 + s; b += (double)(c < d->ctx ? c : d->ctx) * kv_row_bytes; }
    return b;
  };
  int eager_step = 0;
  auto one_step = [&](hipStream_t s_) -> int {
    d->attn_bytes_hint = step_kv_bytes(eager_step);     // K/V bytes ONE layer's attention launch of this step reads (profiler byte counts)
    ++eager_step;
    const DecRows rows{d->row_slot, d->row_pos, d->row_active};
    if (!fused) ETD_TRY(embed(s_));
    float* hf = nullptr;
    ETD_TRY(forward_body(d, n_active, rows, &hf, s_, nullptr, fused, nullptr, nullptr, fused));
    if (fused) {
      const Layer& w0 = d->layers[0];
      DHeadArgs hd = {};
      hd.hfin = hf; hd.M = n_active; hd.H = d->H; hd.V = d->V; hd.Vpad = vpad;
      hd.lnf_g = d->lnfg; hd.lnf_b = d->lnfb; hd.eps = d->cfg.layer_norm_eps; hd.Whead = (const d16*)d->head_frag;
      hd.row_slot = d->row_slot; hd.row_pos = d->row_pos; hd.row_active = d->row_active; hd.row_sp = d->row_sp;
      hd.cur_tok = d->cur_tok; hd.len = d->len; hd.done = d->done; hd.n_out = d->n_out; hd.out_tok = d->out_tok; hd.out_cap = d->out_cap;
      hd.eos = d->eos; hd.limit = d->limit; hd.tgt_attrs = d->tgt_attrs; hd.tgt_proj = d->tgt_proj; hd.tgt_cls = 2; hd.n_bins = d->cfg.num_attribute_bins;
      hd.word = d->word; hd.cls_emb = d->cls_emb; hd.attr_tab = d->attr_tab;
      hd.g1 = w0.ln1g; hd.b1 = w0.ln1b; hd.g2 = w0.ln2g; hd.b2 = w0.ln2b;
      hd.h = d->h; hd.x1 = d->X1b; hd.x2 = d->X2b;
      hd.samp = d->samp_dev; hd.rng_key = d->rng_key;             // the device-side config selects greedy / sampling
      hd.logits_dbg = d->logits_dbg_on ? d->logits_dbg : nullptr;
      ETD_TRY(launch_dstep_head(hd, s_));
      if (d->trace) {
        ETD_TRY(trace_rows(d, d->h, d->H, d->H, 1, 0, n_active, ETD_TRACE_LAYER * d->L * n_active, s_));
        ETD_TRY(trace_rows(d, d->cur_tok, 1, 1, 1, 0, n_active, ETD_TRACE_LAYER * d->L * n_active + n_active, s_));
        hipLaunchKernelGGL(k_trace_next, dim3(1), dim3(1), 0, s_, d->trace_step);
      }
      return ETD_OK;
    }
    ETD_TRY(head_logits(d, hf, n_active, d->logits, s_));
    DArgmaxArgs am = {};
    am.logits = d->logits; am.ldl = d->V; am.V = d->V; am.M = n_active;
    am.rows = rows;
    am.cur_tok = d->cur_tok; am.len = d->len; am.done = d->done; am.n_out = d->n_out; am.out_tok = d->out_tok; am.out_cap = d->out_cap;
    am.eos = d->eos; am.limit = d->limit;
    am.samp = d->samp_dev; am.rng_key = d->rng_key;
    ETD_TRY(launch_dargmax(am, s_));
    return ETD_OK;
  };
  if (fused) {
    // rows, embeddings and first-layer LayerNorms of the FIRST step of this call (every later step gets them from the head kernel)
    ETD_TRY(embed(st));
    const Layer& w0 = d->layers[0];
    ETD_TRY(launch_ln_rows(d->h, n_active, d->H, w0.ln1g, w0.ln1b, w0.ln2g, w0.ln2b, d->cfg.layer_norm_eps, d->X1b, d->X2b, st));
  }
  // The step is ~50 short dependent kernels: replay it as a hipGraph (captured once per n_active; every
  // kernel argument is a fixed workspace/state pointer, the slot list lives in device memory).  Capture needs
  // a non-default stream and must not contain the profiler's event records.
  const bool use_graph = st != nullptr && !prof_enabled() && !ETD_XENV("ETD_NO_GRAPH");
  if (use_graph) {
    const int gkey = 16 * n_active + (d->logits_dbg_on ? 8 : 0) + (d->stamp_on ? 4 : 0) + (d->step_pair ? 2 : 0) + (d->rows_identity ? 1 : 0);
    auto it = d->graphs.find(gkey);
    if (it == d->graphs.end()) {
      hipGraph_t g = nullptr;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = one_step(st);
      const hipError_t ce = hipStreamEndCapture(st, &g);
      eager_step = 0;                                   // (the capture ran one_step once without executing it)
      if (rc != ETD_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (ce != hipSuccess || !g) ETD_FAIL(ETD_EHIP, "decoder_step: stream capture failed: %s", hipGetErrorString(ce));
      hipGraphExec_t ge = nullptr;
      const hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ie != hipSuccess) ETD_FAIL(ETD_EHIP, "decoder_step: graph instantiate failed: %s", hipGetErrorString(ie));
      it = d->graphs.emplace(gkey, ge).first;
    }
    for (int s = 0; s < n_steps; ++s) HIP_TRY(hipGraphLaunch(it->second, st));
  } else {
    for (int s = 0; s < n_steps; ++s) ETD_TRY(one_step(st));
  }
  // exact host-side accounting of what was just issued (graph replays included), then the contexts move on
  {
    double kv = 0;
    for (int s2 = 0; s2 < n_steps; ++s2) kv += step_kv_bytes(s2);
    kv *= d->L;
    d->stat_steps += n_steps; d->stat_row_steps += (double)n_steps * n_active; d->stat_kv_bytes += kv; d->stat_attn_launches += (double)n_steps * d->L;
    if (d->stamp_on) d->stat_stamp_bytes += kv + (fused ? (double)n_steps * d->L * ((double)d->I + d->H) * d->H * 2.0 : 0.0);     // + the down and dense weights each fused launch streams (fp32: the attention launch alone)
    for (int i = 0; i < n_active; ++i) { int& hl = d->host_len[slots[i]]; hl = hl + n_steps < d->ctx - 1 ? hl + n_steps : d->ctx - 1; }
  }
  return ETD_OK;
}

extern "C" int etd_decoder_stats(etd_dec* d, double* out, int n, void* stream) {
  if (!d || !out || n < 1) ETD_FAIL(ETD_EINVAL, "decoder_stats: bad arguments");
  unsigned long long sum_ticks = 0, n_stamped = 0;
  if (d->stamp_dev) {
    std::vector<unsigned long long> sp(ETD_STAMP_HDR);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(sp.data(), d->stamp_dev, ETD_STAMP_HDR * 8, hipMemcpyDeviceToHost));
    sum_ticks = sp[0]; n_stamped = sp[1];
    for (int bank = 0; bank < 2; ++bank) {              // the launch(es) no later launch has folded in yet
      unsigned long long e = 0;
      for (int i = 0; i < 64; ++i) e = std::max(e, sp[8 + 64 * bank + i]);
      if (sp[2 + bank] != 0 && e > sp[2 + bank]) { sum_ticks += e - sp[2 + bank]; ++n_stamped; }
    }
  }
  double w = 0;
  for (const Layer& l : d->layers) w += ((double)l.qkv.N * l.qkv.K + (double)l.dense.N * l.dense.K + (double)l.up.N * l.up.K + (double)l.down.N * l.down.K) * (d->bf16w ? 2.0 : 4.0);
  w += (double)d->V * d->H * (d->bf16w ? 2.0 : 4.0);
  const double v[8] = {d->stat_steps, d->stat_row_steps, d->stat_kv_bytes, d->stat_attn_launches, (double)n_stamped, (double)sum_ticks * 1e-8 /* 100 MHz ticks -> s */, d->stat_stamp_bytes, w};
  for (int i = 0; i < n; ++i) out[i] = i < 8 ? v[i] : 0.0;
  return ETD_OK;
}

// (start, end) of every stamped attention launch since the last reset, in 100 MHz ticks of the device's s_memrealtime (ONE clock for every queue of the chip: the
// logs of several engines can be merged); out_pairs [cap][2]; *n = launches written (the log keeps the first ETD_STAMP_LOGCAP)
extern "C" int etd_decoder_stamp_log(etd_dec* d, unsigned long long* out_pairs, long long cap, long long* n, void* stream) {
  if (!d || !out_pairs || cap < 1 || !n) ETD_FAIL(ETD_EINVAL, "decoder_stamp_log: bad arguments");
  *n = 0;
  if (!d->stamp_dev) return ETD_OK;
  std::vector<unsigned long long> sp(ETD_STAMP_WORDS);
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(hipMemcpy(sp.data(), d->stamp_dev, ETD_STAMP_WORDS * 8, hipMemcpyDeviceToHost));
  long long k = (long long)std::min<unsigned long long>(sp[1], ETD_STAMP_LOGCAP), w = 0;
  for (long long i = 0; i < k && w < cap; ++i, ++w) { out_pairs[2 * w] = sp[ETD_STAMP_HDR + 2 * i]; out_pairs[2 * w + 1] = sp[ETD_STAMP_HDR + 2 * i + 1]; }
  for (int bank = 0; bank < 2 && w < cap; ++bank) {     // the launch(es) no later launch has folded in yet
    unsigned long long e = 0;
    for (int i = 0; i < 64; ++i) e = std::max(e, sp[8 + 64 * bank + i]);
    if (sp[2 + bank] != 0 && e > sp[2 + bank]) { out_pairs[2 * w] = sp[2 + bank]; out_pairs[2 * w + 1] = e; ++w; }
  }
  *n = w;
  return ETD_OK;
}

extern "C" int etd_decoder_stats_reset(etd_dec* d, void* stream) {
  if (!d) ETD_FAIL(ETD_EINVAL, "decoder_stats_reset: null handle");
  d->stat_steps = d->stat_row_steps = d->stat_kv_bytes = d->stat_attn_launches = d->stat_stamp_bytes = 0;
  if (d->stamp_dev) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemset(d->stamp_dev, 0, ETD_STAMP_HDR * 8));          // (the log behind the header is indexed by the launch count, which restarts at zero)
  }
  return ETD_OK;
}

extern "C" int etd_decoder_stamp(etd_dec* d, int on, int skip_steps, void* stream) {
  if (!d || skip_steps < 0) ETD_FAIL(ETD_EINVAL, "decoder_stamp: bad arguments");
  if (on && !d->stamp_dev) {
    HIP_TRY(hipMalloc((void**)&d->stamp_dev, ETD_STAMP_WORDS * 8));
    HIP_TRY(hipMemset(d->stamp_dev, 0, ETD_STAMP_WORDS * 8));
  }
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  d->stamp_armed = on != 0; d->stamp_on = false; d->stamp_skip = on ? skip_steps : 0;
  return ETD_OK;
}

// test hook: pin the attention form of the fused decode step (-1: the host rule; 0: one row per workgroup; 1: two rows of a head per workgroup)
extern "C" int etd_debug_decoder_force_pair(etd_dec* d, int mode) {
  if (!d || mode < -1 || mode > 1) ETD_FAIL(ETD_EINVAL, "force_pair: bad arguments");
  d->force_pair = mode;
  return ETD_OK;
}

// test hook: after etd_debug_decoder_step_logits(d, 1, ...) every decode step stores its logits; out_host (may be null when switching)
// receives the LAST step's [n_active][V] rows, fused d16 step and unfused steps alike
extern "C" int etd_debug_decoder_step_logits(etd_dec* d, int on, float* out_host, int n_active, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || n_active < 0 || n_active > d->S) ETD_FAIL(ETD_EINVAL, "step_logits: bad arguments");
  HIP_TRY(hipStreamSynchronize(st));
  if (out_host && n_active > 0) {
    if (d->last_step_fused && !(d->logits_dbg_on && d->logits_dbg)) ETD_FAIL(ETD_EINVAL, "step_logits: the fused step stores its logits only while the hook is on");
    const float* src = d->last_step_fused ? d->logits_dbg : d->logits;       // (unfused steps leave [n][V] logits in the workspace anyway)
    HIP_TRY(hipMemcpy(out_host, src, (size_t)n_active * d->V * 4, hipMemcpyDeviceToHost));
  }
  if (on && !d->logits_dbg) HIP_TRY(hipMalloc((void**)&d->logits_dbg, (size_t)d->S * d->V * 4));
  d->logits_dbg_on = on != 0;
  return ETD_OK;
}

extern "C" int etd_decoder_poll(etd_dec* d, const int32_t* slots, int n, int32_t* done_out, int32_t* n_out_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !slots || n < 1 || !done_out || !n_out_out) ETD_FAIL(ETD_EINVAL, "decoder_poll: bad args");
  std::vector<int> dnv, nov;
  int *dn, *no;
  if (d->pin_rb) { dn = d->pin_rb; no = d->pin_rb + d->S; }
  else { dnv.resize(d->S); nov.resize(d->S); dn = dnv.data(); no = nov.data(); }
  HIP_TRY(hipMemcpyAsync(dn, d->done, (size_t)d->S * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(no, d->n_out, (size_t)d->S * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) { ETD_TRY(check_slot(d, slots[i])); done_out[i] = dn[slots[i]]; n_out_out[i] = no[slots[i]]; }
  memcpy(d->host_n_out.data(), no, (size_t)d->S * 4);
  d->host_n_out_valid = true;                     // nothing has been launched on this handle's streams since: read_many can size its copy
  return ETD_OK;
}

extern "C" int etd_decoder_read_tokens(etd_dec* d, int slot, int32_t* out, int cap, int* n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ETD_TRY(check_slot(d, slot));
  if (!out || !n) ETD_FAIL(ETD_EINVAL, "read_tokens: null");
  int cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, d->n_out + slot, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (cnt > d->out_cap) cnt = d->out_cap;
  *n = cnt;
  if (cnt > cap) ETD_FAIL(ETD_ENOMEM, "read_tokens: need room for %d tokens", cnt);
  if (cnt > 0) {
    HIP_TRY(hipMemcpyAsync(out, d->out_tok + (size_t)slot * d->out_cap, (size_t)cnt * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return ETD_OK;
}

extern "C" int etd_decoder_read_many(etd_dec* d, int n, const int32_t* slots, int32_t* out, int cap, int32_t* counts, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || n < 1 || !slots || !out || !counts || cap < 1) ETD_FAIL(ETD_EINVAL, "read_many: bad args");
  for (int i = 0; i < n; ++i) ETD_TRY(check_slot(d, slots[i]));
  if (!d->host_n_out_valid) {
    HIP_TRY(hipMemcpyAsync(d->host_n_out.data(), d->n_out, (size_t)d->S * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // only the columns that hold tokens travel: [S rows][widest requested count] out of [S][out_cap]
  int wmax = 0;
  for (int i = 0; i < n; ++i) { int c = d->host_n_out[slots[i]]; if (c > d->out_cap) c = d->out_cap; if (c > wmax) wmax = c; }
  std::vector<int> allv;
  int* all;
  if (d->pin_rb) all = d->pin_rb + 2 * (size_t)d->S; else { allv.resize((size_t)d->S * (wmax > 0 ? wmax : 1)); all = allv.data(); }
  if (wmax > 0) {
    HIP_TRY(hipMemcpy2DAsync(all, (size_t)wmax * 4, d->out_tok, (size_t)d->out_cap * 4, (size_t)wmax * 4, (size_t)d->S, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  for (int i = 0; i < n; ++i) {
    int c = d->host_n_out[slots[i]]; if (c > d->out_cap) c = d->out_cap;
    if (c > cap) ETD_FAIL(ETD_ENOMEM, "read_many: slot %d holds %d tokens (cap %d)", slots[i], c, cap);
    counts[i] = c;
    memcpy(out + (size_t)i * cap, all + (size_t)slots[i] * wmax, (size_t)c * 4);
  }
  return ETD_OK;
}

extern "C" int etd_decoder_generate_bar(etd_dec* d, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                                        const int32_t* tgt_attrs4, int eos_id, int limit, int32_t* out, int* n_out, void* stream) {
  ETD_TRY(etd_decoder_begin_bar(d, slot, ids, cls, attrs4, T, tgt_attrs4, eos_id, limit, stream));
  int dn = 0, no = 0;
  const int32_t sl = slot;
  int chunk = 8;
  while (true) {
    ETD_TRY(etd_decoder_poll(d, &sl, 1, &dn, &no, stream));
    if (dn) break;
    ETD_TRY(etd_decoder_step(d, &sl, 1, chunk, stream));
    if (chunk < 32) chunk *= 2;
  }
  return etd_decoder_read_tokens(d, slot, out, limit, n_out, stream);
}

extern "C" int etd_decoder_prefill_logits(etd_dec* d, int slot, const int32_t* ids, const int32_t* cls, const int32_t* attrs4, int T,
                                          float* logits_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!logits_host) ETD_FAIL(ETD_EINVAL, "prefill_logits: null");
  if (T > d->Mmax) ETD_FAIL(ETD_EINVAL, "prefill_logits: T=%d exceeds %d", T, d->Mmax);
  const int32_t sl = slot, tt = T;
  Staged sg; float* hf = nullptr;
  ETD_TRY(stage_and_forward(d, 1, &sl, &tt, ids, cls, attrs4, nullptr, &sg, &hf, st));
  ETD_TRY(head_logits(d, hf, T, d->logits, st));
  HIP_TRY(hipMemcpyAsync(logits_host, d->logits, (size_t)T * d->V * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}

// diagnostic: a 64-bit sum over every 32-bit word of everything this handle allocated except its weights (KV cache, workspaces,
// stream state) -- changes if anybody, this handle or another, writes a single word of it
__global__ void k_sum_words(const unsigned* __restrict__ p, long long n, unsigned long long* out) {
  unsigned long long s = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += (unsigned long long)p[i] * (unsigned long long)((i & 1023) + 1);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
// out[0] = the sum over everything; out[1 + i] = allocation i of alloc_workspaces (in its order: K cache, V cache, h, h2, Q, AO, DO, M1,
// logits, qkv_raw, hlast, X1b, X2b, AOb, M1b, Pk, row_cnt, Xcat, Qb, samp, rng_key, row_sp, row_slot, ...) while cap allows
extern "C" int etd_debug_decoder_checksum(etd_dec* d, unsigned long long* out, int cap, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !out || cap < 1) ETD_FAIL(ETD_EINVAL, "decoder_checksum: bad arguments");
  const size_t first = d->weights_owner ? 0 : d->n_weight_allocs;
  const size_t n = d->allocs.size() - first;
  unsigned long long* acc = nullptr;
  HIP_TRY(hipMalloc(&acc, 8 * (n + 1)));
  HIP_TRY(hipMemsetAsync(acc, 0, 8 * (n + 1), st));
  for (size_t i = 0; i < n; ++i)
    hipLaunchKernelGGL(k_sum_words, dim3(1024), dim3(256), 0, st, (const unsigned*)d->allocs[first + i], (long long)(d->alloc_bytes[first + i] / 4), acc + 1 + i);
  std::vector<unsigned long long> h(n + 1);
  HIP_TRY(hipMemcpyAsync(h.data(), acc, 8 * (n + 1), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(acc);
  for (size_t i = 1; i <= n; ++i) h[0] += h[i];
  for (int i = 0; i < cap; ++i) out[i] = (size_t)i <= n ? h[i] : 0ull;
  return ETD_OK;
}

// diagnostic: out[(layer * S + slot) * ctx + pos] = 32-bit sum over the K and V words of that position (all heads), d16 caches only
__global__ void k_kv_rowsums(const unsigned* __restrict__ K, const unsigned* __restrict__ V, int L, int S, int nh, int ctx, unsigned* out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)L * S * ctx) return;
  const int pos = (int)(i % ctx), slot = (int)((i / ctx) % S), l = (int)(i / ((long long)ctx * S));
  unsigned s = 0;
  for (int hd = 0; hd < nh; ++hd) {
    const long long w0 = ((((long long)l * S + slot) * nh + hd) * ctx + pos) * 32;      // 64 d16 = 32 words
    for (int w = 0; w < 32; ++w) s += K[w0 + w] * (unsigned)(w + 1 + 64 * hd) + V[w0 + w] * (unsigned)(w + 33 + 64 * hd);
  }
  out[i] = s;
}
extern "C" int etd_debug_decoder_kv_rowsums(etd_dec* d, unsigned* out_host, long long cap, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !out_host || !d->bf16w) ETD_FAIL(ETD_EINVAL, "kv_rowsums: d16 handles only");
  const long long n = (long long)d->L * d->S * d->ctx;
  if (cap < n) ETD_FAIL(ETD_ENOMEM, "kv_rowsums: need room for %lld words", n);
  unsigned* dev = nullptr;
  HIP_TRY(hipMalloc(&dev, n * 4));
  hipLaunchKernelGGL(k_kv_rowsums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const unsigned*)d->Kc, (const unsigned*)d->Vc, d->L, d->S, d->nh, d->ctx, dev);
  HIP_TRY(hipMemcpyAsync(out_host, dev, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(dev);
  return ETD_OK;
}

// Step trace (tools/probe_trace.py): after trace_begin every decode step of the d16 fused path records a hash of each row of each
// kernel's outputs -- per layer Q, gelu(up), the 12 split-K slabs, the residual stream and the next LayerNorm rows; then the next
// step's embeddings and tokens -- into a ring of cap_steps records.  words_per_step = (49 * layers + 2) * n_active.
extern "C" int etd_debug_decoder_trace_begin(etd_dec* d, int cap_steps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || cap_steps < 1 || !d->bf16w) ETD_FAIL(ETD_EINVAL, "trace_begin: bad args");
  HIP_TRY(hipStreamSynchronize(st));
  for (auto& kv : d->graphs) (void)hipGraphExecDestroy(kv.second);       // the captured steps do not hold the trace launches
  d->graphs.clear();
  if (d->trace) { (void)hipFree(d->trace); d->trace = nullptr; }
  if (d->trace_step) { (void)hipFree(d->trace_step); d->trace_step = nullptr; }
  if (d->trace_pk) { (void)hipFree(d->trace_pk); d->trace_pk = nullptr; }
  if (d->trace_q) { (void)hipFree(d->trace_q); d->trace_q = nullptr; }
  HIP_TRY(hipMalloc(&d->trace_pk, (size_t)12 * d->S * d->H * 4));
  HIP_TRY(hipMalloc(&d->trace_q, (size_t)d->S * d->H * 4));
  if (!d->trace_dbg) HIP_TRY(hipMalloc(&d->trace_dbg, (size_t)d->nh * d->S * 256 * 8 * 4));
  const size_t bytes = (size_t)cap_steps * trace_wps(d, d->S) * 4;
  HIP_TRY(hipMalloc(&d->trace, bytes));
  HIP_TRY(hipMalloc(&d->trace_step, 4));
  HIP_TRY(hipMemsetAsync(d->trace, 0, bytes, st));
  HIP_TRY(hipMemsetAsync(d->trace_step, 0, 4, st));
  HIP_TRY(hipStreamSynchronize(st));
  d->trace_cap = cap_steps;
  return ETD_OK;
}
extern "C" int etd_debug_decoder_trace_slabs(etd_dec* d, float* out_host, long long cap_floats, int n_active, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !d->trace_pk || !out_host || n_active < 1 || n_active > d->S) ETD_FAIL(ETD_EINVAL, "trace_slabs: bad args");
  const long long n = 12LL * n_active * d->H;
  if (cap_floats < n) ETD_FAIL(ETD_ENOMEM, "trace_slabs: need room for %lld floats", n);
  HIP_TRY(hipMemcpyAsync(out_host, d->trace_pk, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}
// layer 0's queries [n_active][hidden] of the last traced step, and one (layer, slot, head)'s K and V cache rows [n_pos][64] (d16 bit patterns)
// the attention workgroups' per-lane softmax state of layer 0 in the last traced step: [heads][n_active][256][8]
extern "C" int etd_debug_decoder_trace_lanes(etd_dec* d, float* out_host, long long cap_floats, int n_active, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)(d ? d->nh : 0) * n_active * 256 * 8;
  if (!d || !d->trace_dbg || !out_host || n_active < 1 || n_active > d->S || cap_floats < n) ETD_FAIL(ETD_EINVAL, "trace_lanes: bad args");
  HIP_TRY(hipMemcpyAsync(out_host, d->trace_dbg, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}
extern "C" int etd_debug_decoder_trace_q(etd_dec* d, float* out_host, long long cap_floats, int n_active, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !d->trace_q || !out_host || n_active < 1 || n_active > d->S || cap_floats < (long long)n_active * d->H) ETD_FAIL(ETD_EINVAL, "trace_q: bad args");
  HIP_TRY(hipMemcpyAsync(out_host, d->trace_q, (size_t)n_active * d->H * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}
extern "C" int etd_debug_decoder_peek_kv(etd_dec* d, int layer, int slot, int head, int n_pos, unsigned short* k_out, unsigned short* v_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !d->bf16w || layer < 0 || layer >= d->L || slot < 0 || slot >= d->S || head < 0 || head >= d->nh || n_pos < 1 || n_pos > d->ctx || !k_out || !v_out)
    ETD_FAIL(ETD_EINVAL, "peek_kv: bad args");
  const size_t off = ((size_t)layer * d->layer_stride + (size_t)slot * d->slot_stride + (size_t)head * d->ctx * 64) * 2;
  HIP_TRY(hipMemcpyAsync(k_out, (const char*)d->Kc + off, (size_t)n_pos * 128, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(v_out, (const char*)d->Vc + off, (size_t)n_pos * 128, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}
extern "C" int etd_debug_decoder_trace_read(etd_dec* d, unsigned* out_host, long long cap_words, int n_active, int* steps_done, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!d || !d->trace || !out_host || !steps_done || n_active < 1 || n_active > d->S) ETD_FAIL(ETD_EINVAL, "trace_read: bad args");
  const long long n = (long long)d->trace_cap * trace_wps(d, n_active);
  if (cap_words < n) ETD_FAIL(ETD_ENOMEM, "trace_read: need room for %lld words", n);
  HIP_TRY(hipMemcpyAsync(out_host, d->trace, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(steps_done, d->trace_step, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ETD_OK;
}

extern "C" double etd_decoder_step_bytes(const etd_dec* d, int n_streams, int ctx) {
  // weights once + per stream K and V of `ctx` positions over all layers (+ the appended K/V row)
  const double esz = d->bf16w ? 2.0 : 4.0;
  double w = 0;
  for (const Layer& l : d->layers) w += ((double)l.qkv.N * l.qkv.K + (double)l.dense.N * l.dense.K + (double)l.up.N * l.up.K + (double)l.down.N * l.down.K) * esz;
  w += (double)d->V * d->H * esz;
  const double kv = (double)n_streams * 2.0 * d->L * (double)(ctx + 1) * d->H * esz;
  return w + kv;
}
