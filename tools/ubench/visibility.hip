// Minimal reproducer, independent of the decoder: is a kernel's output always visible to the NEXT kernel of the same stream when
// another stream floods the memory system with scattered sub-line stores?  Stream A: producer / consumer kernel pairs over a 256 KiB
// buffer (the consumer reads elements written by workgroups on other XCDs and counts wrong values).  Stream B (optional): 2-byte
// stores at a 2 KiB stride over 1 GiB (the access pattern of the prefill's V^T scratch), or full-line streaming stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_prod(int* d, int n, int it) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = it * 7 + i; }
// partial-line producer: a 128-byte line (32 ints) is written by TWO workgroups, 64 bytes each, 8 bytes per lane (the K/V append of
// the decode step: two 32-feature tiles = two workgroups, most likely on different XCDs, fill one cache row)
__global__ void k_prod_half(int* d, int n, int it) {
  const int w = blockIdx.x, half = w & 1, t = threadIdx.x;                 // workgroup pair (w >> 1) owns 256 / 8 = 32 lines ... 
  const int line = (w >> 1) * 32 + (t >> 3), e = half * 16 + (t & 7) * 2;   // 8 lanes x 8 bytes = 64 bytes of the line
  const int i = line * 32 + e;
  if (i + 1 < n) { int2 v = {it * 7 + i, it * 7 + i + 1}; *reinterpret_cast<int2*>(d + i) = v; }
}
__global__ void k_cons(const int* d, int n, int it, unsigned long long* err, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = (i + n / 2 + 64 * 5) % n;                       // written by another workgroup, most likely on another XCD
  const int v = d[j];
  if (v != it * 7 + j) atomicAdd(err, 1ull);
  out[i] = v;                                                  // (and something the next producer's stream order depends on)
}
__global__ void k_scatter2(unsigned short* p, long long n_elems, int stride, int iters, unsigned short val) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) { p[(i * stride + it * 37) % n_elems] = val; i += (long long)gridDim.x * blockDim.x; }
}
__global__ void k_stream16(float4* p, long long n, int iters, float v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) { p[i % n] = float4{v, v, v, v}; i += (long long)gridDim.x * blockDim.x; }
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1;               // 0: stream A alone, 1: + scattered 2-byte stores, 2: + streaming 16-byte stores
  const int pairs = argc > 2 ? atoi(argv[2]) : 40000;
  const int n = 64 * 1024;
  int *d, *out; unsigned long long* err; unsigned short* big;
  CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
  const long long big_bytes = 1ll << 30;
  CK(hipMalloc(&big, big_bytes));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  std::atomic<bool> stop{false};
  std::thread tb([&] {
    CK(hipSetDevice(0));
    unsigned short v = 1;
    while (!stop.load() && mode != 0) {
      for (int k = 0; k < 8; ++k) {
        if (mode == 1) hipLaunchKernelGGL(k_scatter2, dim3(2048), dim3(256), 0, sb, big, big_bytes / 2, 1024, 32, v++);
        else hipLaunchKernelGGL(k_stream16, dim3(2048), dim3(256), 0, sb, (float4*)big, big_bytes / 16, 32, (float)v++);
      }
      CK(hipStreamSynchronize(sb));
    }
  });
  for (int it = 1; it <= pairs; ++it) {
    if (argc > 3) hipLaunchKernelGGL(k_prod_half, dim3(n / 32 / 32 * 2), dim3(256), 0, sa, d, n, it);
    else hipLaunchKernelGGL(k_prod, dim3(n / 256), dim3(256), 0, sa, d, n, it);
    hipLaunchKernelGGL(k_cons, dim3(n / 256), dim3(256), 0, sa, d, n, it, err, out);
    if (it % 2000 == 0) CK(hipStreamSynchronize(sa));
  }
  CK(hipStreamSynchronize(sa));
  stop.store(true); tb.join();
  unsigned long long h = 0; CK(hipMemcpy(&h, err, 8, hipMemcpyDeviceToHost));
  printf("mode %d: %d producer/consumer pairs, %llu stale or wrong elements seen by consumers\n", mode, pairs, h);
  return 0;
}
