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
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

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
__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 r = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  return r;
}
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }

__device__ __forceinline__ float xhalf(float v) {  // value held by the lane 32 away
  return __shfl_xor(v, 32, 64);
}
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
#endif
