// How fast can the decode step's attention READ its K/V, and does the layout matter?  Emulates the key loop's access pattern only (no
// softmax): one workgroup of 4 waves per (row, head); a wave-iteration covers 16 keys, 8 lanes x 16 B per 128-byte row, the next iteration's
// requests in flight.  Layout A (the library's): K and V in separate arrays [slot][head][max_ctx][64] bf16.  Layout B: one array
// [slot][head][max_ctx][K 64 | V 64] (a key's K and V rows adjacent: 256 contiguous bytes).  Layout C: as A but every wave reads its
// keys as ONE contiguous block (wave w owns keys [w * ctx/4, (w+1) * ctx/4)).
// build: hipcc -O3 --offload-arch=gfx950 kv_layout.hip -o kv_layout.bin      run: ./kv_layout.bin [rows=54] [ctx=340] [max_ctx=1088] [slots=4*rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ K, const u32x4* __restrict__ V, int rows, int ctx, int max_ctx, int slot0, unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 3, c = lane & 7;
  const int m = blockIdx.x % rows, head = blockIdx.x / rows, slot = slot0 + m;
  // u32x4 units: a 128-byte row = 8 units
  const long long base = ((long long)slot * 8 + head) * max_ctx;
  u32x4 acc = {0, 0, 0, 0};
  auto ld = [&](int key, bool v) -> u32x4 {
    key = key < ctx ? key : ctx - 1;
    if (LAYOUT == 1) return __builtin_nontemporal_load(K + (base + key) * 16 + (v ? 8 : 0) + c);
    return __builtin_nontemporal_load((v ? V : K) + (base + key) * 8 + c);
  };
  if (LAYOUT == 2) {
    const int per = (ctx + 3) / 4, k_lo = wave * per, k_hi = (k_lo + per < ctx ? k_lo + per : ctx);
    u32x4 a0 = ld(k_lo + j, false), b0 = ld(k_lo + j, true), a1 = ld(k_lo + 8 + j, false), b1 = ld(k_lo + 8 + j, true);
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
      u32x4 na0 = a0, nb0 = b0, na1 = a1, nb1 = b1;
      if (k0 + 16 < k_hi) { na0 = ld(k0 + 16 + j, false); nb0 = ld(k0 + 16 + j, true); na1 = ld(k0 + 24 + j, false); nb1 = ld(k0 + 24 + j, true); }
      acc += a0 ^ b0; acc += a1 ^ b1;
      a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
    }
  } else {
    int k0 = wave * 8;
    u32x4 a0 = ld(k0 + j, false), b0 = ld(k0 + j, true), a1 = ld(k0 + 32 + j, false), b1 = ld(k0 + 32 + j, true);
    for (; k0 < ctx; k0 += 64) {
      u32x4 na0 = a0, nb0 = b0, na1 = a1, nb1 = b1;
      if (k0 + 64 < ctx) { na0 = ld(k0 + 64 + j, false); nb0 = ld(k0 + 64 + j, true); na1 = ld(k0 + 96 + j, false); nb1 = ld(k0 + 96 + j, true); }
      acc += a0 ^ b0; acc += a1 ^ b1;
      a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 54, ctx = argc > 2 ? atoi(argv[2]) : 340, max_ctx = argc > 3 ? atoi(argv[3]) : 1088, slots_arg = argc > 4 ? atoi(argv[4]) : 0;
  const int slots = slots_arg > 4 * rows ? slots_arg : 4 * rows;            // four streams x rows slots are touched
  if (ctx > max_ctx || rows < 1) { fprintf(stderr, "bad arguments\n"); return 1; }
  const size_t bytes = (size_t)slots * 8 * max_ctx * 128;                 // one of K, V for `slots` streams of one layer
  const int layers = 8;
  u32x4 *K, *V; unsigned* sink;
  CK(hipMalloc(&K, bytes * 2 * layers)); CK(hipMalloc(&V, bytes * layers)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(K, 1, bytes * 2 * layers)); CK(hipMemset(V, 2, bytes * layers));
  hipStream_t st[4]; for (auto& s : st) CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"A: K and V separate (the library's layout)", "B: K|V of a key adjacent (256-byte rows)", "C: separate, each wave reads one contiguous quarter of the keys"};
  for (int engines = 1; engines <= 4; engines *= 4) {
    for (int lay = 0; lay < 3; ++lay) {
      const int iters = 200;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int it = 0; it < iters; ++it)
          for (int e = 0; e < engines; ++e) {
            const int layer = it % layers;                                   // walk the layers like a step does: the caches never fit any cache
            const u32x4* Kl = K + (size_t)layer * (lay == 1 ? bytes * 2 : bytes) / 16; const u32x4* Vl = V + (size_t)layer * bytes / 16;
            if (lay == 0) hipLaunchKernelGGL(k_read<0>, dim3(rows * 8), dim3(256), 0, st[e], Kl, Vl, rows, ctx, max_ctx, e * rows, sink);
            else if (lay == 1) hipLaunchKernelGGL(k_read<1>, dim3(rows * 8), dim3(256), 0, st[e], Kl, Vl, rows, ctx, max_ctx, e * rows, sink);
            else hipLaunchKernelGGL(k_read<2>, dim3(rows * 8), dim3(256), 0, st[e], Kl, Vl, rows, ctx, max_ctx, e * rows, sink);
          }
        for (int e = 1; e < engines; ++e) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[e])); CK(hipStreamWaitEvent(st[0], ev, 0)); CK(hipEventDestroy(ev)); }
        CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double by = (double)rows * 8 * ctx * 256.0 * engines;
      printf("%d stream(s) x %d rows x 8 heads x ctx %d, layout %s: %.2f us per launch round, %.2f TB/s\n", engines, rows, ctx, names[lay], 1e3 * best / iters, by * iters / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
