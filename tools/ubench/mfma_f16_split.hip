// Stand-alone check (hipcc -O3 --offload-arch=gfx950): can v_mfma_f32_32x32x16_f16 carry an fp32 GEMM as a two-term f16 split?
//   (1) do f16 SUBNORMAL A/B inputs come through un-flushed?   (2) error of hi*hi + hi*lo + lo*hi (+ lo*lo) against an fp64 product,
//   next to a k-ordered fp32 fmaf chain (= what v_mfma_f32_32x32x2_f32 computes) and next to the same split in bf16.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A [32][K] row-major planes, B [32][K] (B^T: column n's K values contiguous), C [32][32]
template <int TERMS, bool BF>
__global__ __launch_bounds__(64) void k_split(const unsigned short* Ah, const unsigned short* Al, const unsigned short* Bh, const unsigned short* Bl, int K, float* C) {
  const int lane = threadIdx.x, r = lane & 31, kq = lane >> 5;
  f16v acc = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int off = r * K + k0 + kq * 8;
    if constexpr (!BF) {
      h8 ah = *(const h8*)(Ah + off), al = *(const h8*)(Al + off), bh = *(const h8*)(Bh + off), bl = *(const h8*)(Bl + off);
      if (TERMS >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
      if (TERMS >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      if (TERMS >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    } else {
      b8 ah = *(const b8*)(Ah + off), al = *(const b8*)(Al + off), bh = *(const b8*)(Bh + off), bl = *(const b8*)(Bl + off);
      if (TERMS >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
      if (TERMS >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      if (TERMS >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i >> 2) * 8 + kq * 4 + (i & 3)) * 32 + r] = acc[i];   // C[m = A row][n = B row]: A rows on the register index, B rows on the lane
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float b2f(unsigned short s) { unsigned u = (unsigned)s << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int K = 512;
  std::vector<float> A(32 * K), B(32 * K);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  unsigned short *dAh, *dAl, *dBh, *dBl; float* dC;
  hipMalloc(&dAh, 32 * K * 2); hipMalloc(&dAl, 32 * K * 2); hipMalloc(&dBh, 32 * K * 2); hipMalloc(&dBl, 32 * K * 2); hipMalloc(&dC, 32 * 32 * 4);
  std::vector<unsigned short> Ah(32 * K), Al(32 * K), Bh(32 * K), Bl(32 * K);
  std::vector<float> C(32 * 32);
  // ---- (1) subnormal inputs: A = 2^-20 (f16 subnormal), B = 1 -> every C = K * 2^-20 when not flushed
  for (int i = 0; i < 32 * K; ++i) { Ah[i] = f2h(ldexpf(1.f, -20)); Al[i] = 0; Bh[i] = f2h(1.f); Bl[i] = 0; }
  hipMemcpy(dAh, Ah.data(), 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dAl, Al.data(), 32 * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(dBh, Bh.data(), 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dBl, Bl.data(), 32 * K * 2, hipMemcpyHostToDevice);
  k_split<1, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC);
  hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 1.0, K=%d: C[0] = %g (expected %g) -> f16 subnormal inputs %s\n", K, C[0], K * ldexp(1.0, -20), C[0] == (float)(K * ldexp(1.0, -20)) ? "PRESERVED" : "FLUSHED");
  for (int i = 0; i < 32 * K; ++i) { Ah[i] = f2h(1.f); Bh[i] = f2h(ldexpf(1.f, -22)); }
  hipMemcpy(dAh, Ah.data(), 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dBh, Bh.data(), 32 * K * 2, hipMemcpyHostToDevice);
  k_split<1, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC);
  hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
  printf("1.0 x subnormal B (2^-22): C[0] = %g (expected %g) -> %s\n", C[0], K * ldexp(1.0, -22), C[0] == (float)(K * ldexp(1.0, -22)) ? "PRESERVED" : "FLUSHED");
  // ---- (2) accuracy on three operand distributions
  for (int dist = 0; dist < 3; ++dist) {
    for (int i = 0; i < 32 * K; ++i) {
      float a = rnd(), b = rnd();
      if (dist == 1) { a = a * expf(4.f * rnd()); b = 0.05f * b * expf(3.f * rnd()); }          // wide dynamic range: activations x weights
      if (dist == 2) { a = fabsf(a) < 0.7f ? a * 1e-3f : a * 4.f; b *= 0.03f; }                 // GELU-like: mostly tiny, a few large
      A[i] = a; B[i] = b;
    }
    std::vector<double> R(32 * 32); std::vector<float> F(32 * 32); double ssum = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      double s = 0, sa = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * B[n * K + k]; sa += fabs((double)A[m * K + k] * B[n * K + k]); f = fmaf(A[m * K + k], B[n * K + k], f); }
      R[m * 32 + n] = s; F[m * 32 + n] = f; ssum += sa;
    }
    const double sabs = ssum / 1024;           // mean of sum |a b| : the scale errors are quoted against
    double e32 = 0; for (int i = 0; i < 1024; ++i) e32 = fmax(e32, fabs(F[i] - R[i]));
    printf("dist %d: sum|ab| = %.3g; fp32 fmaf chain: max err %.3g (%.3g of sum|ab|)\n", dist, sabs, e32, e32 / sabs);
    for (int bf = 0; bf < 2; ++bf) {
      // f16: operands scaled by a power of two so that max |x| sits at 2^14 (exact; undone on the result)
      float ma = 0, mb = 0; for (int i = 0; i < 32 * K; ++i) { ma = fmaxf(ma, fabsf(A[i])); mb = fmaxf(mb, fabsf(B[i])); }
      int ea, eb; frexpf(ma, &ea); frexpf(mb, &eb);
      const float sa = bf ? 1.f : ldexpf(1.f, 15 - ea), sb = bf ? 1.f : ldexpf(1.f, 15 - eb);
      for (int i = 0; i < 32 * K; ++i) {
        if (!bf) { Ah[i] = f2h(A[i] * sa); Al[i] = f2h(A[i] * sa - h2f(Ah[i])); Bh[i] = f2h(B[i] * sb); Bl[i] = f2h(B[i] * sb - h2f(Bh[i])); }
        else { Ah[i] = f2b(A[i]); Al[i] = f2b(A[i] - b2f(Ah[i])); Bh[i] = f2b(B[i]); Bl[i] = f2b(B[i] - b2f(Bh[i])); }
      }
      hipMemcpy(dAh, Ah.data(), 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dAl, Al.data(), 32 * K * 2, hipMemcpyHostToDevice);
      hipMemcpy(dBh, Bh.data(), 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dBl, Bl.data(), 32 * K * 2, hipMemcpyHostToDevice);
      for (int terms = 1; terms <= 4; ++terms) {
        if (bf) { if (terms == 1) k_split<1, true><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); else if (terms == 2) k_split<2, true><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC);
                  else if (terms == 3) k_split<3, true><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); else k_split<4, true><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); }
        else { if (terms == 1) k_split<1, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); else if (terms == 2) k_split<2, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC);
               else if (terms == 3) k_split<3, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); else k_split<4, false><<<1, 64>>>(dAh, dAl, dBh, dBl, K, dC); }
        hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
        double e = 0, rms = 0; for (int i = 0; i < 1024; ++i) { double d = C[i] / ((double)sa * sb) - R[i]; e = fmax(e, fabs(d)); rms += d * d; }
        printf("   %s split, %d MFMA term(s): max err %.3g (%.3g of sum|ab|), rms %.3g\n", bf ? "bf16" : "f16 ", terms, e, e / sabs, sqrt(rms / 1024));
      }
    }
  }
  return 0;
}
