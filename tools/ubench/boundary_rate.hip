// How many dependent kernel boundaries per microsecond does the GPU sustain, on 1, 2 and 4 queues?  Each stream replays a hipGraph of 25
// dependent launches (a decode step's shape); the kernels are empty (1 workgroup), or 64 workgroups that each dirty 16 KiB (so that the
// release at the boundary has lines to write back), or 432 workgroups reading 37 MB (an attention launch's stream).
// build: hipcc -O3 --offload-arch=gfx950 boundary_rate.hip -o boundary_rate.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(256) void k_dirty(float* p, float v) { float4* q = reinterpret_cast<float4*>(p) + ((size_t)blockIdx.x * 1024 + threadIdx.x); for (int i = 0; i < 4; ++i) q[i * 256] = float4{v, v, v, v}; }
__global__ __launch_bounds__(256) void k_read(const float4* p, float* sink, size_t per_wg) {
  const float4* q = p + (size_t)blockIdx.x * per_wg; float4 a = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i < per_wg; i += 256) { const float4 t = q[i]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
  if (a.x + a.y + a.z + a.w == 1.2345f) sink[0] = a.x;
}
int main() {
  float* buf; CK(hipMalloc(&buf, (size_t)1 << 30)); CK(hipMemset(buf, 0, (size_t)1 << 30));
  float* sink; CK(hipMalloc(&sink, 64));
  hipStream_t st[4]; hipGraphExec_t ge[3][4];
  for (int q = 0; q < 4; ++q) CK(hipStreamCreate(&st[q]));
  const char* names[3] = {"empty kernels (1 workgroup)", "64 workgroups dirtying 16 KiB each", "432 workgroups reading 37 MB (streams of 87 KiB)"};
  for (int kind = 0; kind < 3; ++kind)
    for (int q = 0; q < 4; ++q) {
      hipGraph_t g; CK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 25; ++i) {
        if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[q], (int*)nullptr);
        else if (kind == 1) hipLaunchKernelGGL(k_dirty, dim3(64), dim3(256), 0, st[q], buf + (size_t)q * (64 << 20) / 4 + (size_t)(i % 8) * (2 << 20) / 4, (float)i);
        else hipLaunchKernelGGL(k_read, dim3(432), dim3(256), 0, st[q], reinterpret_cast<const float4*>(buf) + (size_t)q * (160 << 20) / 16 + (size_t)(i % 4) * (40 << 20) / 16, sink, (size_t)87 * 1024 / 16);
      }
      CK(hipStreamEndCapture(st[q], &g)); CK(hipGraphInstantiate(&ge[kind][q], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    }
  for (int kind = 0; kind < 3; ++kind)
    for (int Q = 1; Q <= 4; Q *= 2) {
      const int reps = kind == 2 ? 100 : 400;
      double best = 1e9;
      for (int t = 0; t < 3; ++t) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) for (int q = 0; q < Q; ++q) CK(hipGraphLaunch(ge[kind][q], st[q]));
        CK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (s < best) best = s;
      }
      printf("%-52s %d queue(s): %.2f us per launch on a queue, %.3f launches per us in all\n", names[kind], Q, 1e6 * best / (reps * 25), Q * reps * 25 / (1e6 * best));
    }
  return 0;
}
