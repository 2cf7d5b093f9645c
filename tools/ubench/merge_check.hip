// Stand-alone check of the instruction form behind LABNOTES.md "packed FP32 with crossed op_sel": the three-stage softmax merge of
// the decode step's attention core, once as plain C++ (hipcc's SLP vectoriser pairs lr's sum with o[0]'s crosswise: v_pk_mul_f32 x2 +
// v_pk_add_f32 / v_pk_fma_f32 whose LOW result reads a HIGH register) and once with every sum as scalar VALU instructions of its own
// (inline asm: mul, mul, add).  Same inputs, same roundings: the two must agree bit for bit.  256 threads, ~10 KiB of LDS, like the
// attention workgroups.  Run alone and beside another stream's MFMA kernels (tools/probe_merge.py).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC merge_check.hip -o libmerge_check.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ unsigned mixu(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u ^ (d + 1u) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float unitf(unsigned h) { return __uint_as_float(0x3f800000u | (h >> 9)); }      // [1, 2)
__device__ __forceinline__ float sum_scalar(float a, float b, float c, float d) {
  float t, u, r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u) : "v"(c), "v"(d));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(u));
  return r;
}
#define MERGE_STAGE(off, SUM)                                                                                         \
  {                                                                                                                   \
    const float m2 = __shfl_xor(mr, off, 64), l2 = __shfl_xor(lr, off, 64);                                           \
    const float mn = fmaxf(mr, m2);                                                                                   \
    const float f1 = (mr == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mr - mn), f2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn); \
    lr = SUM(lr, f1, l2, f2);                                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) { const float o2 = __shfl_xor(o[e], off, 64); o[e] = SUM(o[e], f1, o2, f2); } \
    mr = mn;                                                                                                          \
  }
#define SUM_PLAIN(a, b, c, d) ((a) * (b) + (c) * (d))

template <bool SCALAR>
__device__ __forceinline__ void merge(float& mr, float& lr, float (&o)[8]) {
  if constexpr (SCALAR) { MERGE_STAGE(8, sum_scalar) MERGE_STAGE(16, sum_scalar) MERGE_STAGE(32, sum_scalar) }
  else { MERGE_STAGE(8, SUM_PLAIN) MERGE_STAGE(16, SUM_PLAIN) MERGE_STAGE(32, SUM_PLAIN) }
}

// errs[0]: lanes whose lr differs between the two versions; errs[1]: lanes whose o[] differs; lanes[64]: differing results per lane
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_merge_check(int iters, const float* __restrict__ gsrc, long long gn,
                                                                                            unsigned long long* errs, unsigned long long* lanes, unsigned* first) {
  __shared__ unsigned pad[2432];
  const int tid = threadIdx.x, lane = tid & 63;
  pad[(tid * 9 + iters) & 2047] = tid;
  unsigned e0 = 0, e1 = 0;
  for (int it = 0; it < iters; ++it) {
    // loads in flight across the merge, as the dense-weight fragments are in the real kernel
    const float4 g = *reinterpret_cast<const float4*>(gsrc + ((long long)(mixu(blockIdx.x, tid, it, 9) % (unsigned)(gn / 4 - 1))) * 4);
    float mr0 = unitf(mixu(blockIdx.x, tid, it, 0)) * 8.f - 12.f, lr0 = unitf(mixu(blockIdx.x, tid, it, 1)) * 5.f, o0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o0[e] = unitf(mixu(blockIdx.x, tid, it, 2 + e)) - 1.5f;
    float mrA = mr0, lrA = lr0, oA[8], mrB = mr0, lrB = lr0, oB[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { oA[e] = o0[e]; oB[e] = o0[e]; }
    merge<false>(mrA, lrA, oA);
    merge<true>(mrB, lrB, oB);
    bool od = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) od |= __float_as_uint(oA[e]) != __float_as_uint(oB[e]);
    if (__float_as_uint(lrA) != __float_as_uint(lrB)) {
      ++e0;
      if (atomicCAS(first + 7, 0u, 1u) == 0u) { first[0] = blockIdx.x; first[1] = tid; first[2] = it; first[3] = __float_as_uint(lrA); first[4] = __float_as_uint(lrB); }
    }
    if (od) ++e1;
    if (g.x + g.y + g.z + g.w == 12345.678f) ++e1;
  }
  if (e0) atomicAdd(errs + 0, (unsigned long long)e0);
  if (e1) atomicAdd(errs + 1, (unsigned long long)e1);
  if (e0 + e1) atomicAdd(lanes + lane, (unsigned long long)(e0 + e1));
  __syncthreads();
  if (pad[(tid * 5) & 2047] == 0x1234567u) errs[7] = 1;
}
extern "C" int merge_check(int n_wg, int iters, const float* gsrc, long long gn, unsigned long long* errs, unsigned long long* lanes, unsigned* first, void* stream) {
  hipLaunchKernelGGL(k_merge_check, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, iters, gsrc, gn, errs, lanes, first);
  return (int)hipGetLastError();
}
