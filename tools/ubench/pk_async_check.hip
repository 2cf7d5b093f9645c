// Crossed packed-FP32 ops (v_pk_mul_f32 x2 + v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], the form of LABNOTES.md) issued WHILE
// asynchronous VGPR writers of the same wave are in flight: eight ds_bpermute_b32 results (mode bit 0) and/or four global_load_dwordx4
// results (mode bit 1) land in other registers around the packed sequence.  Every packed result is compared with scalar v_mul_f32 /
// v_add_f32 on the same inputs; the exchanged / loaded values are checked too.  Run beside another stream's MFMA kernels
// (tools/probe_pk.py with PROBE_PK_ASYNC=mode).  build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC pk_async_check.hip -o libpk_async_check.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mixu(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u ^ (d + 1u) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float unitf(unsigned h) { return __uint_as_float(0x3f800000u | (h >> 9)); }

// errs[0]: packed results wrong; errs[1]: exchanged values wrong; errs[2]: loaded values wrong; lanes[64]: packed errors per lane
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_pk_async_check(int iters, int mode, const u32x4_t* __restrict__ gsrc, long long gq,
                                                                                               unsigned long long* errs, unsigned long long* lanes, unsigned* first) {
  __shared__ unsigned pad[2432];
  const int tid = threadIdx.x, lane = tid & 63;
  pad[(tid * 9 + iters) & 2047] = tid;
  unsigned e0 = 0, e1 = 0, e2 = 0;
  const int addr = ((lane ^ 8) << 2);
  for (int it = 0; it < iters; ++it) {
    const unsigned h = mixu(blockIdx.x, tid, it, 0);
    f32x2_t f = {unitf(h), unitf(h * 3u + 1u)};                      // (f2, f1)
    f32x2_t p = {unitf(h * 5u + 2u), unitf(h * 7u + 3u)};            // (o2_a, o_b)
    f32x2_t q = {unitf(h * 11u + 4u), unitf(h * 13u + 5u)};          // (o2_b, o_a)
    unsigned s[8], d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = mixu(blockIdx.x, tid, it, 10 + k); d[k] = 0; }
    const long long gi = (long long)(mixu(blockIdx.x, tid, it, 99) % (unsigned)(gq - 4));
    u32x4_t g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
    const u32x4_t* gp = gsrc + gi;
    f32x2_t x;
    if (mode == 1) {
      asm volatile(
          "ds_bpermute_b32 %0, %10, %11\n\tds_bpermute_b32 %1, %10, %12\n\tds_bpermute_b32 %2, %10, %13\n\tds_bpermute_b32 %3, %10, %14\n\t"
          "ds_bpermute_b32 %4, %10, %15\n\tds_bpermute_b32 %5, %10, %16\n\tds_bpermute_b32 %6, %10, %17\n\tds_bpermute_b32 %7, %10, %18\n\t"
          "v_pk_mul_f32 %8, %8, %19\n\t"
          "v_pk_mul_f32 %9, %9, %19\n\t"
          "v_pk_add_f32 %8, %8, %9 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+v"(p), "+v"(q)
          : "v"(addr), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(f));
      x = p;
    } else if (mode == 2 || mode >= 16) {
      // mode 2: loads, then the packed sequence at once.  mode 16 + v: variants of the same with the loads in flight --
      //   v = 1: s_nop 0 between, 2: s_nop 3, 3: s_nop 7 x2, 4: final add NOT crossed (control), 5: only the crossed add (operands multiplied by scalar ops before the loads),
      //   6: the packed sequence BEFORE the loads are issued (nothing in flight), 7: s_waitcnt vmcnt(0) before the packed sequence
#define PKA_LOADS "global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\tglobal_load_dwordx4 %2, %6, off offset:32\n\tglobal_load_dwordx4 %3, %6, off offset:48\n\t"
#define PKA_MULS "v_pk_mul_f32 %4, %4, %7\n\tv_pk_mul_f32 %5, %5, %7\n\t"
#define PKA_CROSS "v_pk_add_f32 %4, %4, %5 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define PKA_PLAIN "v_pk_add_f32 %4, %4, %5\n\t"
#define PKA_OUT : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "+v"(p), "+v"(q) : "v"(gp), "v"(f)
      const int v = mode >= 16 ? mode - 16 : 0;
      if (v == 0) asm volatile(PKA_LOADS PKA_MULS PKA_CROSS "s_waitcnt vmcnt(0)" PKA_OUT);
      else if (v == 1) asm volatile(PKA_LOADS "s_nop 0\n\t" PKA_MULS PKA_CROSS "s_waitcnt vmcnt(0)" PKA_OUT);
      else if (v == 2) asm volatile(PKA_LOADS "s_nop 3\n\t" PKA_MULS PKA_CROSS "s_waitcnt vmcnt(0)" PKA_OUT);
      else if (v == 3) asm volatile(PKA_LOADS "s_nop 7\n\ts_nop 7\n\t" PKA_MULS PKA_CROSS "s_waitcnt vmcnt(0)" PKA_OUT);
      else if (v == 4) asm volatile(PKA_LOADS PKA_MULS PKA_PLAIN "s_waitcnt vmcnt(0)" PKA_OUT);
      else if (v == 5) {
        p[0] *= f[0]; p[1] *= f[1]; q[0] *= f[0]; q[1] *= f[1];
        asm volatile("" : "+v"(p), "+v"(q));
        asm volatile(PKA_LOADS PKA_CROSS "s_waitcnt vmcnt(0)" PKA_OUT);
      } else if (v == 6) asm volatile(PKA_MULS PKA_CROSS PKA_LOADS "s_waitcnt vmcnt(0)" PKA_OUT);
      else asm volatile(PKA_LOADS "s_waitcnt vmcnt(0)\n\t" PKA_MULS PKA_CROSS PKA_OUT);
      x = p;
      if (v == 4) {          // control: the expected values of the uncrossed add
        const float pl = unitf(h * 5u + 2u), ph = unitf(h * 7u + 3u), ql = unitf(h * 11u + 4u), qh = unitf(h * 13u + 5u);
        if (__float_as_uint(x[0]) != __float_as_uint(pl * f[0] + ql * f[0]) || __float_as_uint(x[1]) != __float_as_uint(ph * f[1] + qh * f[1])) ++e0;
        if (g0.x != mixu((unsigned)(gi * 4), 1, 2, 3)) ++e2;
        continue;
      }
    } else {
      asm volatile(
          "global_load_dwordx4 %0, %14, off\n\tglobal_load_dwordx4 %1, %14, off offset:16\n\tglobal_load_dwordx4 %2, %14, off offset:32\n\tglobal_load_dwordx4 %3, %14, off offset:48\n\t"
          "ds_bpermute_b32 %4, %15, %16\n\tds_bpermute_b32 %5, %15, %17\n\tds_bpermute_b32 %6, %15, %18\n\tds_bpermute_b32 %7, %15, %19\n\t"
          "ds_bpermute_b32 %8, %15, %20\n\tds_bpermute_b32 %9, %15, %21\n\tds_bpermute_b32 %10, %15, %22\n\tds_bpermute_b32 %11, %15, %23\n\t"
          "v_pk_mul_f32 %12, %12, %24\n\t"
          "v_pk_mul_f32 %13, %13, %24\n\t"
          "v_pk_add_f32 %12, %12, %13 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+v"(p), "+v"(q)
          : "v"(gp), "v"(addr), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(f));
      x = p;
    }
    // expected: lo = p.lo * f.lo + q.hi * f.hi ; hi = p.hi * f.hi + q.lo * f.lo   (inputs recomputed from the hash: p, q were overwritten)
    const float pl = unitf(h * 5u + 2u), ph = unitf(h * 7u + 3u), ql = unitf(h * 11u + 4u), qh = unitf(h * 13u + 5u);
    float t0, t1, u0, u1, r0, r1;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(pl), "v"(f[0]));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(ph), "v"(f[1]));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u0) : "v"(ql), "v"(f[0]));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u1) : "v"(qh), "v"(f[1]));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(t0), "v"(u1));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(t1), "v"(u0));
    if (__float_as_uint(x[0]) != __float_as_uint(r0) || __float_as_uint(x[1]) != __float_as_uint(r1)) {
      ++e0;
      if (atomicCAS(first + 7, 0u, 1u) == 0u) { first[0] = blockIdx.x; first[1] = tid; first[2] = it; first[3] = __float_as_uint(x[0]); first[4] = __float_as_uint(r0); first[5] = __float_as_uint(x[1]); first[6] = __float_as_uint(r1); }
    }
    if (mode < 16 && (mode & 1)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) if (d[k] != mixu(blockIdx.x, tid ^ 8, it, 10 + k)) ++e1;
    }
    if (mode >= 16 || (mode & 2)) {
      const unsigned w = (unsigned)(gi * 4);
      if (g0.x != mixu(w, 1, 2, 3) || g1.x != mixu(w + 4, 1, 2, 3) || g2.y != mixu(w + 9, 1, 2, 3) || g3.w != mixu(w + 15, 1, 2, 3)) ++e2;
    }
  }
  if (e0) { atomicAdd(errs + 0, (unsigned long long)e0); atomicAdd(lanes + lane, (unsigned long long)e0); }
  if (e1) atomicAdd(errs + 1, (unsigned long long)e1);
  if (e2) atomicAdd(errs + 2, (unsigned long long)e2);
  __syncthreads();
  if (pad[(tid * 5) & 2047] == 0x1234567u) errs[7] = 1;
}
__global__ void k_fill(unsigned* dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = mixu((unsigned)i, 1, 2, 3);
}
extern "C" int pk_async_fill(unsigned* dst, long long n, void* stream) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, (hipStream_t)stream, dst, n); return (int)hipGetLastError(); }
extern "C" int pk_async_check(int n_wg, int iters, int mode, const void* gsrc, long long gq, unsigned long long* errs, unsigned long long* lanes, unsigned* first, void* stream) {
  hipLaunchKernelGGL(k_pk_async_check, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, iters, mode, (const u32x4_t*)gsrc, gq, errs, lanes, first);
  return (int)hipGetLastError();
}
