#include "prof.h"

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/etude_hip_debug.h"
#include "ext_kernels.h"
#include "dec_kernels.h"

namespace {
struct Entry { double ms = 0; long long n = 0; double flops = 0, bytes = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
std::mutex g_mu;
bool g_on = false;
std::map<std::string, Entry> g_ent;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool prof_enabled() { return g_on; }
bool launch_skipped(const char* name) {
  static const char* only = getenv("ETD_EXT_ONLY");
  if (!only || !*only) return false;
  if (!strcmp(name, "k_linear_dec") || !strcmp(name, "k_attn_causal") || !strcmp(name, "k_pqkv") || !strcmp(name, "k_pattn")) return false;     // the decoder's batched prefill, never the Extract stage
  const size_t n = strlen(name);
  for (const char* p = only; *p;) {
    const char* q = strchr(p, ',');
    const size_t len = q ? (size_t)(q - p) : strlen(p);
    if (len == n && !strncmp(p, name, n)) return false;
    p += len + (q ? 1 : 0);
  }
  return true;
}

bool launch_log_on() { static const bool on = ETD_XENV("ETD_LAUNCH_LOG") && atoi(ETD_XENV("ETD_LAUNCH_LOG")) > 0; return on; }
void launch_log(const char* name, hipStream_t st, bool after) {
  if (!after) { fprintf(stderr, "[launch] %s\n", name); fflush(stderr); return; }
  const hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { fprintf(stderr, "[launch] %s FAILED: %s\n", name, hipGetErrorString(e)); fflush(stderr); }
}

hipEvent_t prof_begin(hipStream_t st) {
  if (!g_on) return nullptr;
  hipEvent_t e;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    e = get_event();
  }
  if (e) (void)hipEventRecord(e, st);
  return e;
}
void prof_end(const char* name, hipEvent_t start, hipStream_t st, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t stop = get_event();
  if (!stop) { g_pool.push_back(start); return; }
  (void)hipEventRecord(stop, st);
  Entry& en = g_ent[name];
  en.pending.push_back({start, stop});
  en.n += 1; en.flops += flops; en.bytes += bytes;
}

extern "C" int etd_prof_enable(int on) { std::lock_guard<std::mutex> lk(g_mu); g_on = on != 0; return ETD_OK; }
extern "C" int etd_prof_collect(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_ent) {
    for (auto& p : kv.second.pending) {
      float ms = 0.f;
      if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) kv.second.ms += ms;
      g_pool.push_back(p.first); g_pool.push_back(p.second);
    }
    kv.second.pending.clear();
  }
  return ETD_OK;
}
extern "C" int etd_prof_reset(void) {
  etd_prof_collect();
  std::lock_guard<std::mutex> lk(g_mu);
  g_ent.clear();
  return ETD_OK;
}
extern "C" int etd_prof_count(void) { std::lock_guard<std::mutex> lk(g_mu); return (int)g_ent.size(); }
extern "C" int etd_prof_entry(int i, char* name, int name_cap, double* total_ms, long long* launches, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (i < 0 || i >= (int)g_ent.size() || !name || name_cap < 2) ETD_FAIL(ETD_EINVAL, "prof_entry: bad index");
  auto it = g_ent.begin();
  std::advance(it, i);
  strncpy(name, it->first.c_str(), name_cap - 1); name[name_cap - 1] = 0;
  if (total_ms) *total_ms = it->second.ms;
  if (launches) *launches = it->second.n;
  if (flops) *flops = it->second.flops;
  if (bytes) *bytes = it->second.bytes;
  return ETD_OK;
}

// ---- measurement hook: cost of a dependent kernel boundary on this stack (eager vs hipGraph replay)
struct BigArg { int v[120]; };
__global__ void k_empty_small(int* p) { if (p && threadIdx.x == 1024) p[0] = 1; }
__global__ void k_empty_big(BigArg a, int* p) { if (p && threadIdx.x == 1024) p[0] = a.v[3]; }
extern "C" int etd_debug_boundary_cost(int n_nodes, int iters, int big_args, void* stream, double* eager_us, double* graph_us) {
  hipStream_t st = (hipStream_t)stream;
  if (!st || n_nodes < 1 || iters < 1 || !eager_us || !graph_us) ETD_FAIL(ETD_EINVAL, "boundary_cost: need a non-default stream");
  BigArg ba = {};
  auto body = [&]() {
    for (int i = 0; i < n_nodes; ++i) {
      if (big_args) hipLaunchKernelGGL(k_empty_big, dim3(64), dim3(256), 0, st, ba, (int*)nullptr);
      else hipLaunchKernelGGL(k_empty_small, dim3(64), dim3(256), 0, st, (int*)nullptr);
    }
  };
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  body(); HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventRecord(e0, st));
  for (int it = 0; it < iters; ++it) body();
  HIP_TRY(hipEventRecord(e1, st)); HIP_TRY(hipEventSynchronize(e1));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *eager_us = 1e3 * ms / ((double)iters * n_nodes);
  hipGraph_t g; hipGraphExec_t ge;
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  body();
  HIP_TRY(hipStreamEndCapture(st, &g));
  HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  HIP_TRY(hipGraphLaunch(ge, st)); HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventRecord(e0, st));
  for (int it = 0; it < iters; ++it) HIP_TRY(hipGraphLaunch(ge, st));
  HIP_TRY(hipEventRecord(e1, st)); HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *graph_us = 1e3 * ms / ((double)iters * n_nodes);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ETD_OK;
}

// ---- measurement hook: the token-major GEMM (k_linear, plain bf16 epilogue) on a synthetic [M,K] x [N,K]^T problem.
// tools/bench_linear.py sweeps the shapes the extractor and the decoder prefill use; with ETD_LIN_STAMP=n in the
// environment the first n launches print the in-kernel phase stamps instead of being timed.
__global__ void k_fill_bf16(bf16* p, long long n, unsigned seed) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (bf16)(((int)(x & 0xffff) - 32768) * (1.f / 262144.f));
  }
}
extern "C" int etd_debug_linear(int M, int N, int K, int iters, void* stream, double* us) {
  hipStream_t st = (hipStream_t)stream;
  if (M < 1 || N % 256 || K % 128 || iters < 1 || !us) ETD_FAIL(ETD_EINVAL, "debug_linear: bad shape");
  bf16 *X = nullptr, *W = nullptr, *Y = nullptr; float* b = nullptr;
  HIP_TRY(hipMalloc(&X, (size_t)M * K * 2 + 256)); HIP_TRY(hipMalloc(&W, (size_t)N * K * 2 + 256));
  HIP_TRY(hipMalloc(&Y, (size_t)M * N * 2 + 256)); HIP_TRY(hipMalloc(&b, (size_t)N * 4));
  HIP_TRY(hipMemsetAsync(b, 0, (size_t)N * 4, st));
  hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, st, X, (long long)M * K, 1u);
  hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, st, W, (long long)N * K, 2u);      // (random values: the layout does not matter for timing)
  LinArgs a = {};
  a.X = (const e16*)X; a.ldx = K; a.W = (const e16*)W; a.bias = b; a.M = M; a.N = N; a.K = K; a.Y = (e16*)Y; a.ldy = N; a.vt_block = -1;      // (timing only: 16-bit patterns of either type)
  int rc = launch_linear(a, 1, st);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventRecord(e0, st));
  for (int it = 0; it < iters && rc == ETD_OK; ++it) rc = launch_linear(a, 1, st);
  HIP_TRY(hipEventRecord(e1, st)); HIP_TRY(hipEventSynchronize(e1));
  float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *us = 1e3 * ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(b);
  return rc;
}

// Diagnostic aggressors (tools/probe_race.py): `iters` launches of ONE kernel type on private buffers filled with random values.
//   which 0: k_attn, 2048 sequences x 256 queries x 256 keys, 4 heads (the extractor's shape)
//         1: k_attn causal, 54 ragged sequences of ~340 rows, 8 heads (the decoder prefill's shape)
//         2: k_linear<2> (LayerNorm epilogue), M = 131072, K = 256
//         3: k_ln_rows, 18432 rows x 512
extern "C" int etd_debug_kernel_loop(int which, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (which < 0 || which > 3 || iters < 1) ETD_FAIL(ETD_EINVAL, "debug_kernel_loop: bad arguments");
  std::vector<void*> bufs;
  auto get = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes + 256) != hipSuccess) return nullptr; bufs.push_back(p); return p; };
  auto fill = [&](void* p, size_t n_bf16, unsigned seed) { hipLaunchKernelGGL(k_fill_bf16, dim3(1024), dim3(256), 0, st, (bf16*)p, (long long)n_bf16, seed); };
  int rc = ETD_OK;
  if (which == 0 || which == 1) {
    const int nh = which == 0 ? 4 : 8, H = nh * 64;
    const int n_seq = which == 0 ? 2048 : 54, S = which == 0 ? 256 : 340, Spad = which == 0 ? 256 : 384;
    const long long M = (long long)n_seq * S;
    bf16 *Q = (bf16*)get(M * H * 2), *K = (bf16*)get(M * H * 2), *VT = (bf16*)get((size_t)n_seq * nh * 64 * Spad * 2), *O = (bf16*)get(M * H * 2);
    int* meta = (int*)get(n_seq * 8);
    if (!Q || !K || !VT || !O || !meta) rc = ETD_ENOMEM;
    if (rc == ETD_OK) {
      fill(Q, M * H, 1u); fill(K, M * H, 2u); fill(VT, (size_t)n_seq * nh * 64 * Spad, 3u);
      std::vector<int> hm(2 * n_seq);
      for (int i = 0; i < n_seq; ++i) { hm[i] = i * S; hm[n_seq + i] = S; }
      HIP_TRY(hipMemcpyAsync(meta, hm.data(), hm.size() * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      AttnArgs a = {};
      a.Q = (const e16*)Q; a.ldq = H; a.q_seq_stride = (long long)S * H; a.K = (const e16*)K; a.ldk = H; a.k_seq_stride = (long long)S * H; a.VT = (const e16*)VT; a.Spad = Spad;
      a.O = (e16*)O; a.ldo = H; a.o_seq_stride = (long long)S * H; a.n_seq = n_seq; a.Sq = S; a.Sk = S; a.scale_log2e = 0.125f * 1.4426950408889634f; a.n_heads = nh;
      if (which == 1) { a.seq_row0 = meta; a.seq_len = meta + n_seq; a.causal = 1; }
      for (int it = 0; it < iters && rc == ETD_OK; ++it) rc = launch_attn(a, st);
    }
  } else if (which == 2) {
    const int M = 131072, K = 256, N = 256;
    bf16 *X = (bf16*)get((size_t)M * K * 2), *W = (bf16*)get((size_t)N * K * 2), *R = (bf16*)get((size_t)M * N * 2), *Y = (bf16*)get((size_t)M * N * 2);
    float* par = (float*)get(3 * N * 4);
    if (!X || !W || !R || !Y || !par) rc = ETD_ENOMEM;
    if (rc == ETD_OK) {
      fill(X, (size_t)M * K, 1u); fill(W, (size_t)N * K, 2u); fill(R, (size_t)M * N, 3u);
      std::vector<float> hp(3 * N, 0.f);
      for (int i = 0; i < N; ++i) hp[N + i] = 1.f;
      HIP_TRY(hipMemcpyAsync(par, hp.data(), hp.size() * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      LinArgs a = {};
      a.X = (const e16*)X; a.ldx = K; a.W = (const e16*)W; a.bias = par; a.M = M; a.N = N; a.K = K; a.Y = (e16*)Y; a.ldy = N; a.vt_block = -1; a.R = (const e16*)R; a.ldr = N; a.gamma = par + N; a.beta = par + 2 * N;
      for (int it = 0; it < iters && rc == ETD_OK; ++it) rc = launch_linear_ln(a, st);
    }
  } else {
    const int M = 18432, H = 512;
    float* h = (float*)get((size_t)M * H * 4); float* par = (float*)get(4 * H * 4);
    d16 *x1 = (d16*)get((size_t)M * H * 2), *x2 = (d16*)get((size_t)M * H * 2);
    if (!h || !par || !x1 || !x2) rc = ETD_ENOMEM;
    if (rc == ETD_OK) {
      fill(h, (size_t)M * H * 2, 7u);            // (random bf16 pairs read as floats: finite, any magnitude)
      HIP_TRY(hipMemsetAsync(par, 0, 4 * H * 4, st));
      for (int it = 0; it < iters && rc == ETD_OK; ++it) rc = launch_ln_rows(h, M, H, par, par + H, par + 2 * H, par + 3 * H, 1e-5f, x1, x2, st);
    }
  }
  (void)hipStreamSynchronize(st);
  for (void* p : bufs) (void)hipFree(p);
  return rc;
}

// Diagnostic: an empty kernel with k_embed's footprint (82 KiB of static LDS, 296 registers) launched from INSIDE this library
__global__ __launch_bounds__(256) void k_dbg_empty(int* sink, int never) {
  __shared__ int big[81728 / 4];
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a39, v255" ::: "v255", "a39");
  if (never) { for (int i = threadIdx.x; i < 81728 / 4; i += 256) big[i] = i; __syncthreads(); sink[threadIdx.x] = big[(threadIdx.x * 7) % (81728 / 4)]; }
}
extern "C" int etd_debug_empty_launch(int gx, int gy, int gz, int* sink, void* stream) {
  if (gx < 1 || gy < 1 || gz < 1) ETD_FAIL(ETD_EINVAL, "empty_launch: bad grid");
  hipLaunchKernelGGL(k_dbg_empty, dim3(gx, gy, gz), dim3(256), 0, (hipStream_t)stream, sink, 0);
  HIP_TRY(hipGetLastError());
  return ETD_OK;
}
