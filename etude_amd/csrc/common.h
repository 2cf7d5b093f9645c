// Shared device/host helpers for the etude_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// ---- measurement switches.  The shipped library reads the environment only for the switches a test or a tool of this tree sets (getenv at their call sites); the A/B and
// diagnostic switches of earlier rounds' experiments (LABNOTES.md names them) are read only in a -DETD_EXPERIMENTS build (`etd_has_experiments()`), and are absent otherwise.
#include <stdlib.h>
#ifdef ETD_EXPERIMENTS
#define ETD_XENV(name) getenv(name)
#else
#define ETD_XENV(name) ((const char*)nullptr)
#endif

// ---- error plumbing (no exceptions cross the C ABI) -------------------------------------------
extern thread_local std::string g_etd_err;
#define ETD_OK 0
#define ETD_EINVAL (-22)
#define ETD_ENOMEM (-12)
#define ETD_EHIP (-5)
#define ETD_FAIL(code, ...)                                   \
  do {                                                        \
    char _b[512];                                             \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                    \
    g_etd_err = _b;                                           \
    return (code);                                            \
  } while (0)
#define HIP_TRY(x)                                                                            \
  do {                                                                                        \
    hipError_t _e = (x);                                                                      \
    if (_e != hipSuccess) ETD_FAIL(ETD_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(_e)); \
  } while (0)
#define ETD_TRY(x)            \
  do {                        \
    int _r = (x);             \
    if (_r != ETD_OK) return _r; \
  } while (0)

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  // v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k 8*(l>>5)+j], B[k 8*(l>>5)+j][col l&31];
  // D[row (i&3)+8*(i>>2)+4*(l>>5)][col l&31] in register i.
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {      // v_mfma_f32_32x32x16_f16: the same layout and rate, 11 significant bits per operand instead of 8
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 r = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  return r;
}
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ float bf2f(f16 x) { return (float)x; }
__device__ __forceinline__ f16x4 pack4h(float a, float b, float c, float d) {
  f16x4 r = {(f16)a, (f16)b, (f16)c, (f16)d};
  return r;
}

// ---- cross-lane exchange without the LDS crossbar.  lane_xor<O>(v) is the value lane (id ^ O) holds -- what
// __shfl_xor(v, O, 64) returns, but hipcc lowers that to ds_bpermute_b32 (an LDS-pipe round trip, ~100 clk of latency per
// stage in a dependent reduction chain).  Within a row of 16 lanes DPP does it inside the VALU; across rows gfx950 has
// v_permlane16_swap / v_permlane32_swap.  All lanes of the wave must be active (wave-uniform control flow), as for any reduction.
__device__ __forceinline__ int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int O>
__device__ __forceinline__ unsigned lane_xor_u(unsigned v) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "lane_xor: power of two below 64");
  if constexpr (O == 1) return __builtin_amdgcn_update_dpp(0u, v, 0xb1, 0xf, 0xf, false);         // quad_perm [1,0,3,2]
  else if constexpr (O == 2) return __builtin_amdgcn_update_dpp(0u, v, 0x4e, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
  else if constexpr (O == 4) {
    unsigned o = __builtin_amdgcn_update_dpp(0u, v, 0x124, 0xf, 0xa, false);                       // row_ror:4  -> banks 1,3 take lane i-4
    return __builtin_amdgcn_update_dpp(o, v, 0x12c, 0xf, 0x5, false);                              // row_ror:12 -> banks 0,2 take lane i+4
  } else if constexpr (O == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false); // row_ror:8
  else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (wave_lane() & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (wave_lane() & 32) ? r[0] : r[1];
  }
}
template <int O> __device__ __forceinline__ float lane_xor(float v) { return __uint_as_float(lane_xor_u<O>(__float_as_uint(v))); }
template <int O> __device__ __forceinline__ int lane_xor(int v) { return (int)lane_xor_u<O>((unsigned)v); }

__device__ __forceinline__ float xhalf(float v) {  // value held by the lane 32 away
  return __shfl_xor(v, 32, 64);
}
// Butterfly reductions, stages 32, 16, 8, 4, 2, 1: every lane ends with the total, and the order of additions is fixed.
// Two formulations with bit-identical results:
//  * wave_sum / wave_max: __shfl_xor (ds_bpermute).  The exchange runs in the LDS pipe, off the VALU -- the right choice for
//    kernels with many waves per CU whose VALU is the busy unit (measured: k_dattn's key loop is 6 % SLOWER with DPP).
//  * wave_sum_lat / wave_max_lat: lane_xor (DPP / permlane swap).  ~10x shorter dependent-chain latency -- for kernels that
//    run a handful of waves and wait on every stage (k_dstep_head).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_lat(float v) {
  v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
  return v;
}
__device__ __forceinline__ float wave_max_lat(float v) {
  v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
  v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
  return v;
}
// N independent wave_sum_lat's (same additions in the same order per value); the chains interleave.
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = wave_sum_lat(v[j]);
}
#endif
