// Stand-alone reproducer (no library, no Python): on MI355X a packed-FP32 add whose op_sel swaps the halves of an operand
//     v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]        (D.lo = A.lo + B.hi, D.hi = A.hi + B.lo)
// returns D.lo = A.lo (as if B.hi were 0) in lanes 48-63, now and then, while another stream's MFMA kernel shares the SIMD.  The
// uncrossed add on the same operands never fails; alone the crossed add never fails.  LABNOTES.md, "packed FP32 with crossed op_sel".
//   build: hipcc -O3 --offload-arch=gfx950 pk_cross_repro.hip -o pk_cross_repro.bin        run: ./pk_cross_repro.bin [seconds per case = 2]
// Victim: 432 workgroups x 256 threads, ~10 KiB LDS, occupancy 5 (the decode step's attention workgroups).  Aggressors, one at a time on a
// second stream: 0 none, 1 register-only MFMA loop, 2 MFMA fed by ds_read_b128 from a 64 KiB LDS ring, 3 the same + global_load_lds
// refills of the ring, 4 register-only MFMA with ~250 live registers (two waves fill a SIMD's register file), 5 the same with 15
// accumulators, 6 fourteen accumulators fed from the LDS ring.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mixu(unsigned a, unsigned b, unsigned c, unsigned d) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u ^ (d + 1u) * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float unitf(unsigned h) { return __uint_as_float(0x3f800000u | (h >> 9)); }

// errs[k]: wrong results of form k (see FORMS in main); lanes[64]: errors of form 0 per lane
#define PK2(name, text, lo, hi) { f32x2_t r_; asm volatile(text : "=v"(r_) : "v"(a), "v"(b)); float e0_, e1_; lo; hi; \
                                  if (__float_as_uint(r_[0]) != __float_as_uint(e0_) || __float_as_uint(r_[1]) != __float_as_uint(e1_)) ++name; }
#define PK3(name, text, lo, hi) { f32x2_t r_; asm volatile(text : "=v"(r_) : "v"(a), "v"(b), "v"(c)); float e0_, e1_; lo; hi; \
                                  if (__float_as_uint(r_[0]) != __float_as_uint(e0_) || __float_as_uint(r_[1]) != __float_as_uint(e1_)) ++name; }
#define ADD(d, x, y) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define MUL(d, x, y) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define FMA(d, x, y, z) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_victim(int iters, unsigned long long* errs, unsigned long long* lanes) {
  __shared__ unsigned pad[2432];
  const int tid = threadIdx.x, lane = tid & 63;
  pad[(tid * 9 + iters) & 2047] = tid;
  unsigned e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned h = mixu(blockIdx.x, tid, it, 0);
    const f32x2_t a = {unitf(h), unitf(h * 3u + 1u)}, b = {unitf(h * 5u + 2u), unitf(h * 7u + 3u)}, c = {unitf(h * 11u + 4u), unitf(h * 13u + 5u)};
    PK2(e0, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", ADD(e0_, a[0], b[1]), ADD(e1_, a[1], b[0]))
    PK2(e1, "v_pk_add_f32 %0, %1, %2", ADD(e0_, a[0], b[0]), ADD(e1_, a[1], b[1]))
    PK2(e2, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", ADD(e0_, a[1], b[0]), ADD(e1_, a[0], b[1]))
    PK2(e3, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", MUL(e0_, a[0], b[1]), MUL(e1_, a[1], b[0]))
    PK3(e4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]", FMA(e0_, a[0], b[0], c[1]), FMA(e1_, a[1], b[1], c[0]))
    PK3(e5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", FMA(e0_, a[0], b[1], c[0]), FMA(e1_, a[1], b[1], c[1]))
    PK2(e6, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", ADD(e0_, a[0], b[1]), ADD(e1_, a[1], b[1]))
    PK2(e7, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", MUL(e0_, a[0], b[0]), MUL(e1_, a[1], b[0]))
  }
  if (e0) { atomicAdd(errs + 0, (unsigned long long)e0); atomicAdd(lanes + lane, (unsigned long long)e0); }
  if (e1) atomicAdd(errs + 1, (unsigned long long)e1);
  if (e2) atomicAdd(errs + 2, (unsigned long long)e2);
  if (e3) atomicAdd(errs + 3, (unsigned long long)e3);
  if (e4) atomicAdd(errs + 4, (unsigned long long)e4);
  if (e5) atomicAdd(errs + 5, (unsigned long long)e5);
  if (e6) atomicAdd(errs + 6, (unsigned long long)e6);
  if (e7) atomicAdd(errs + 7, (unsigned long long)e7);
  __syncthreads();
  if (pad[(tid * 5) & 2047] == 0x1234567u) errs[8] = 1;
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void k_aggr(int iters, const bf16x8_t* __restrict__ gsrc, float* sink) {
  __shared__ __attribute__((aligned(16))) bf16x8_t ring[KIND == 2 || KIND == 3 || KIND == 6 ? 4096 : 1];       // 64 KiB
  const int tid = threadIdx.x;
  unsigned x = (tid + blockIdx.x * 256u) * 2654435761u + 17u;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { x ^= x >> 13; x *= 2246822519u; a[j] = (__bf16)(((int)(x & 0xff) - 128) * (1.f / 256.f)); x ^= x >> 15; b[j] = (__bf16)(((int)(x & 0xff) - 128) * (1.f / 256.f)); }
  constexpr int NACC = KIND == 4 || KIND == 6 ? 14 : (KIND == 5 ? 15 : 4);
  f32x16_t c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = f32x16_t{};
  if constexpr (KIND == 2 || KIND == 3 || KIND == 6) {
    for (int i = tid; i < 4096; i += 256) ring[i] = gsrc[(blockIdx.x * 4096 + i) & 0xfffff];
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 2 || KIND == 3 || KIND == 6) {
      a = ring[(it * 256 + tid) & 4095]; b = ring[(it * 256 + tid + 2048) & 4095];
      if constexpr (KIND == 3) {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + ((blockIdx.x * 4096 + it * 256 + tid) & 0xfffff)), (lptr_t)(ring + ((it * 256 + 1024) & 4095) + (tid & ~63)), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a : b, (i & 2) ? a : b, c[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? b : a, (i & 2) ? b : a, c[i], 0, 0, 0);
    if constexpr (KIND == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) s += c[i][k];
  if (s == 1.2345678f) sink[0] = s;
}

template <bool X_IS_A>
__global__ __launch_bounds__(256, 2) void k_aggr_proj(int iters, const bf16x8_t* __restrict__ gsrc, float* sink) {
  __shared__ __attribute__((aligned(16))) bf16x8_t ring[4096];       // 64 KiB of "weights"
  const int tid = threadIdx.x, lane = tid & 63;
  bf16x8_t xf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) xf[s] = gsrc[(blockIdx.x * 4096 + s * 256 + tid) & 0xfffff];
  for (int i = tid; i < 4096; i += 256) ring[i] = gsrc[(blockIdx.x * 4096 + i) & 0xfffff];
  __syncthreads();
  f32x16_t acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = f32x16_t{};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bf16x8_t w = ring[((it * 32 + c * 8 + t) * 64 + lane) & 4095];
        if constexpr (X_IS_A) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[4 * c + (t >> 1)], w, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xf[4 * c + (t >> 1)], acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) for (int k = 0; k < 16; ++k) s += acc[t][k];
  if (s == 1.2345678f) sink[0] = s;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  unsigned long long *errs, *lanes; float* sink; bf16x8_t* gsrc;
  CK(hipMalloc(&errs, 128)); CK(hipMalloc(&lanes, 512)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&gsrc, (size_t)(1 << 20) * 16));
  CK(hipMemset(gsrc, 0x3c, (size_t)(1 << 20) * 16));
  hipStream_t sv, sa; CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sa));
  const char* names[9] = {"no aggressor", "register-only MFMA loop", "MFMA fed by ds_read_b128 from a 64 KiB LDS ring", "the same + global_load_lds refills", "register-only MFMA, ~250 live registers", "register-only MFMA, 15 accumulators", "MFMA with 14 accumulators fed by ds_read_b128 from a 64 KiB LDS ring",
                          "projection shape: 16 resident token fragments as B, weights from LDS as A, 8 accumulators", "projection shape with the token fragments as the A operand"};
  const bool all = argc > 2;
  for (int kind = 0; kind < 9; ++kind) {
    if (!all && kind != 0 && kind != 1 && kind != 7 && kind != 8) continue;
    std::atomic<bool> stop{false};
    std::thread th([&] {
      if (kind == 0) return;
      CK(hipSetDevice(0));
      while (!stop.load()) {
        for (int r = 0; r < 4; ++r) {
          if (kind == 1) hipLaunchKernelGGL(k_aggr<1>, dim3(1024), dim3(256), 0, sa, 400, gsrc, sink);
          else if (kind == 2) hipLaunchKernelGGL(k_aggr<2>, dim3(1024), dim3(256), 0, sa, 400, gsrc, sink);
          else if (kind == 3) hipLaunchKernelGGL(k_aggr<3>, dim3(1024), dim3(256), 0, sa, 400, gsrc, sink);
          else if (kind == 4) hipLaunchKernelGGL(k_aggr<4>, dim3(1024), dim3(256), 0, sa, 120, gsrc, sink);
          else if (kind == 5) hipLaunchKernelGGL(k_aggr<5>, dim3(1024), dim3(256), 0, sa, 120, gsrc, sink);
          else if (kind == 6) hipLaunchKernelGGL(k_aggr<6>, dim3(1024), dim3(256), 0, sa, 120, gsrc, sink);
          else if (kind == 7) hipLaunchKernelGGL(k_aggr_proj<false>, dim3(1024), dim3(256), 0, sa, 60, gsrc, sink);
          else hipLaunchKernelGGL(k_aggr_proj<true>, dim3(1024), dim3(256), 0, sa, 60, gsrc, sink);
        }
        CK(hipStreamSynchronize(sa));
      }
    });
    CK(hipMemsetAsync(errs, 0, 128, sv)); CK(hipMemsetAsync(lanes, 0, 512, sv)); CK(hipStreamSynchronize(sv));
    const auto t0 = std::chrono::steady_clock::now();
    long long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_victim, dim3(432), dim3(256), 0, sv, 40, errs, lanes);
      CK(hipStreamSynchronize(sv)); n += 20;
    }
    stop.store(true); th.join(); CK(hipDeviceSynchronize());
    unsigned long long he[16], hl[64];
    CK(hipMemcpy(he, errs, 128, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl, lanes, 512, hipMemcpyDeviceToHost));
    unsigned long long g[8] = {0};
    for (int i = 0; i < 64; ++i) g[i >> 3] += hl[i];
    static const char* FORMS[8] = {"add op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)", "add (plain)", "add op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped)",
                                   "mul op_sel:[0,1] op_sel_hi:[1,0]", "fma op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 halves swapped)", "fma op_sel:[0,1,0] (src1 high broadcast)",
                                   "add op_sel:[0,1] (src1 high broadcast)", "mul op_sel_hi:[1,0] (src1 low broadcast)"};
    printf("aggressor %d (%s): %lld victim launches x 432 workgroups x 256 lanes x 40 iterations\n", kind, names[kind], n);
    for (int k = 0; k < 8; ++k) printf("    v_pk_%-62s wrong %llu\n", FORMS[k], he[k]);
    printf("    form 0 errors by lanes 0-7, 8-15, ...: %llu %llu %llu %llu %llu %llu %llu %llu\n", g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
    fflush(stdout);
  }
  return 0;
}
