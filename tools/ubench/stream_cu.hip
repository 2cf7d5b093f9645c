// How fast can a SUBSET of the CUs stream HBM?  (design question: could a decoder engine confined to a quarter of the chip by a CU
// mask still pull its K/V stream at a useful share of the HBM peak?)  G workgroups (one per CU: 96 KiB of dynamic LDS each), W waves
// per workgroup, each wave walks its own contiguous span once with D 1-KiB wave-loads in flight, by register loads (nontemporal
// 16 B per lane) or by LDS-DMA (global_load_lds into a private ring).  Prints GB/s of the launch and per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int D>
__global__ void k_reg(const u32x4* __restrict__ src, long long span_v /* u32x4 per wave */, u32x4* sink) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const u32x4* p = src + ((long long)blockIdx.x * nw + wave) * span_v + lane;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[D];
  const long long n = span_v / 64;       // wave-loads
#pragma unroll
  for (int i = 0; i < D; ++i) buf[i] = __builtin_nontemporal_load(p + (long long)i * 64);
  for (long long i = D; i < n; i += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      acc ^= buf[j];
      buf[j] = __builtin_nontemporal_load(p + (i + j) * 64);
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j) acc ^= buf[j];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc;
  if (threadIdx.x == 9999) lds[0] = 1;
}

template <int D>
__global__ void k_dma(const u32x4* __restrict__ src, long long span_v, u32x4* sink) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const u32x4* p = src + ((long long)blockIdx.x * nw + wave) * span_v + lane;
  unsigned char* ring = lds + wave * (D * 1024);
  const long long n = span_v / 64;
  for (long long i = 0; i < n; i += D) {
#pragma unroll
    for (int j = 0; j < D; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(p + (i + j) * 64), (lptr_t)(ring + j * 1024), 16, 0, 0);
    // keep D / 2 .. D pieces in flight: wait until at most D / 2 are outstanding before reusing the first half of the ring
    if (D >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D / 2) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ring[lane] == 0x5a && lane == 77) sink[0] = u32x4{1, 2, 3, 4};
}

int main(int argc, char** argv) {
  const size_t total = (size_t)3 << 30;        // 3 GiB source
  u32x4 *src, *sink;
  CK(hipMalloc(&src, total)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(src, 1, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int Gs[] = {32, 64, 128, 256};
  const int Ws[] = {4, 8, 16};
  auto run = [&](const char* name, auto kern, int D, int G, int W, size_t ldsb) {
    // every wave streams `span` bytes; the launch moves ~1.5 GiB (less with few CUs: at least 2 MiB per wave)
    size_t span = ((size_t)1536 << 20) / ((size_t)G * W);
    if (span > ((size_t)16 << 20)) span = (size_t)16 << 20;
    span = span / (1024 * 16) * (1024 * 16);
    const long long span_v = span / 16;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(G), dim3(64 * W), ldsb, 0, src, span_v, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double bytes = (double)span * G * W;
    printf("%-4s D=%2d G=%3d W=%2d  %7.1f GB/s  (%6.1f GB/s per CU, %5.0f KiB in flight per CU)\n", name, D, G, W, bytes / best / 1e6, bytes / best / 1e6 / G, (double)D * W);
    fflush(stdout);
  };
  for (int G : Gs)
    for (int W : Ws) {
      const size_t big = 96 * 1024;     // one workgroup per CU
      run("reg", k_reg<2>, 2, G, W, big); run("reg", k_reg<4>, 4, G, W, big); run("reg", k_reg<8>, 8, G, W, big); run("reg", k_reg<16>, 16, G, W, big);
      if (W * 4 * 1024 <= (int)big) run("dma", k_dma<4>, 4, G, W, big);
      if (W * 8 * 1024 <= 160 * 1024) run("dma", k_dma<8>, 8, G, W, W * 8 * 1024 > (int)big ? W * 8 * 1024 : big);
      if (W * 16 * 1024 <= 160 * 1024) run("dma", k_dma<16>, 16, G, W, W * 16 * 1024 > (int)big ? W * 16 * 1024 : big);
    }
  return 0;
}
