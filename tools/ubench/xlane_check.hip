// Self-checking cross-lane / LDS exchanges, shaped like the attention workgroups of the decode step (256 threads, ~10 KiB of LDS,
// about 70 registers): every lane's value is a hash of (workgroup, lane, iteration), so what an exchange must return is known
// without any communication.  Run beside another stream's kernels (tools/probe_xlane.py); errs[] counts wrong results per test:
//   errs[0] __shfl_xor butterflies (ds_bpermute_b32), ten values in flight like the softmax merge
//   errs[1] the same exchanges through DPP / v_permlane swaps
//   errs[2] LDS: 8 lanes per wave write 10 words, barrier, wave 0 reads every wave's words (the cross-wave merge)
//   errs[3] global loads kept in flight across the exchanges arrive intact
// first[0..7]: details of the first ds_bpermute error seen (workgroup, lane, iteration, offset, got, expected, value index, 1)
#include <hip/hip_runtime.h>
#include <cstdint>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mixu(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u ^ (d + 1u) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
template <int O> __device__ __forceinline__ unsigned dpp_xor(unsigned v) {
  if constexpr (O == 1) return __builtin_amdgcn_update_dpp(0u, v, 0xb1, 0xf, 0xf, false);
  else if constexpr (O == 2) return __builtin_amdgcn_update_dpp(0u, v, 0x4e, 0xf, 0xf, false);
  else if constexpr (O == 4) { unsigned o = __builtin_amdgcn_update_dpp(0u, v, 0x124, 0xf, 0xa, false); return __builtin_amdgcn_update_dpp(o, v, 0x12c, 0xf, 0x5, false); }
  else if constexpr (O == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false);
  else if constexpr (O == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? r[0] : r[1]; }
  else { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? r[0] : r[1]; }
}

__global__ __launch_bounds__(256) void k_xlane_check(int iters, int mode, const u32x4_t* __restrict__ gsrc, long long gwords, unsigned long long* errs, unsigned* first) {
  __shared__ unsigned red[4][8][10];
  __shared__ unsigned pad[2048 + 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 3, c = lane & 7;
  const unsigned wg = blockIdx.x;
  unsigned e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  pad[(tid * 9 + iters) & 2047] = tid;                 // (keeps the allocation at the attention workgroup's ~10 KiB)
  for (int it = 0; it < iters; ++it) {
    // global loads in flight across the exchanges (mode & 8)
    u32x4_t ga = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
    long long gi = 0;
    if (mode & 8) {
      gi = (long long)(mixu(wg, tid, it, 77) % (unsigned)(gwords / 4 - 256));
      ga = __builtin_nontemporal_load(gsrc + gi); gb = __builtin_nontemporal_load(gsrc + gi + 128);
    }
    unsigned v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = mixu(wg, lane + 64 * wave, it, k);
    if (mode & 1) {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        unsigned x[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) x[k] = __shfl_xor(v[k], off, 64);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const unsigned want = mixu(wg, (lane ^ off) + 64 * wave, it, k);
          if (x[k] != want) {
            ++e0;
            if (atomicCAS(first + 7, 0u, 1u) == 0u) { first[0] = wg; first[1] = tid; first[2] = it; first[3] = off; first[4] = x[k]; first[5] = want; first[6] = k; }
          }
        }
      }
    }
    if (mode & 2) {
#define XL_STAGE(O) { unsigned x[10]; _Pragma("unroll") for (int k = 0; k < 10; ++k) x[k] = dpp_xor<O>(v[k]); \
                      _Pragma("unroll") for (int k = 0; k < 10; ++k) if (x[k] != mixu(wg, (lane ^ O) + 64 * wave, it, k)) ++e1; }
      XL_STAGE(1) XL_STAGE(2) XL_STAGE(4) XL_STAGE(8) XL_STAGE(16) XL_STAGE(32)
#undef XL_STAGE
    }
    if (mode & 4) {
      if (j == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) red[wave][c][k] = v[k];
      }
      __syncthreads();
      if (wave == 0 && j == 0) {
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int k = 0; k < 10; ++k) if (red[w][c][k] != mixu(wg, c + 64 * w, it, k)) ++e2;
      }
      __syncthreads();
    }
    if (mode & 8) {
      // the source buffer holds word i = mixu(i, 1, 2, 3)
      const unsigned w0 = (unsigned)(gi * 4), w1 = (unsigned)((gi + 128) * 4);
      if (ga.x != mixu(w0, 1, 2, 3) || ga.y != mixu(w0 + 1, 1, 2, 3) || ga.z != mixu(w0 + 2, 1, 2, 3) || ga.w != mixu(w0 + 3, 1, 2, 3)) ++e3;
      if (gb.x != mixu(w1, 1, 2, 3) || gb.y != mixu(w1 + 1, 1, 2, 3) || gb.z != mixu(w1 + 2, 1, 2, 3) || gb.w != mixu(w1 + 3, 1, 2, 3)) ++e3;
    }
  }
  if (e0) atomicAdd(errs + 0, (unsigned long long)e0);
  if (e1) atomicAdd(errs + 1, (unsigned long long)e1);
  if (e2) atomicAdd(errs + 2, (unsigned long long)e2);
  if (e3) atomicAdd(errs + 3, (unsigned long long)e3);
  __syncthreads();
  if (pad[(tid * 5 + mode) & 2047] == 0x1234567u) errs[7] = 1;
}
__global__ void k_xlane_fill(unsigned* dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = mixu((unsigned)i, 1, 2, 3);
}
extern "C" int xlane_fill(unsigned* dst, long long n, void* stream) {
  hipLaunchKernelGGL(k_xlane_fill, dim3(4096), dim3(256), 0, (hipStream_t)stream, dst, n);
  return (int)hipGetLastError();
}
extern "C" int xlane_check(int n_wg, int iters, int mode, const void* gsrc, long long gwords, unsigned long long* errs, unsigned* first, void* stream) {
  hipLaunchKernelGGL(k_xlane_check, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, iters, mode, (const u32x4_t*)gsrc, gwords, errs, first);
  return (int)hipGetLastError();
}
