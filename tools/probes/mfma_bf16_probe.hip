// Probe: verifies the operand layout assumed for v_mfma_f32_32x32x16_bf16 on gfx950 with random (asymmetric) A and B:
//   A[i][k]: lane l holds i = l & 31, k = 8*(l >> 5) + e (e = 0..7);  B[k][j]: lane l holds j = l & 31, same k;
//   D[i][j]: col j = l & 31, row i = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5).
// Also checks the bf16 hi/lo split (a = hi + lo) used by the bf16x3 kernels.  Build: hipcc --offload-arch=gfx950 -O2 -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short f2bf(float x) {   // round-to-nearest-even
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ void probe(const float* A, const float* B, float* D, float* D3) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  union { bf16x8 v; unsigned short s[8]; } ah, al, bh, bl;
  for (int e = 0; e < 8; ++e) {
    const float a = A[i * 16 + kb * 8 + e], b = B[(kb * 8 + e) * 32 + i];
    ah.s[e] = f2bf(a); al.s[e] = f2bf(a - bf2f(ah.s[e]));
    bh.s[e] = f2bf(b); bl.s[e] = f2bf(b - bf2f(bh.s[e]));
  }
  f32x16 acc = {0}, acc3 = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc, 0, 0, 0);
  acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, acc3, 0, 0, 0);
  acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, acc3, 0, 0, 0);
  acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc3, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
    D[row * 32 + i] = acc[r];
    D3[row * 32 + i] = acc3[r];
  }
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), D3(32 * 32);
  srand(1);
  for (auto& x : A) x = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto& x : B) x = (rand() / (float)RAND_MAX) * 2 - 1;
  float *dA, *dB, *dD, *dD3;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&dD3, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD3);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(D3.data(), dD3, D.size() * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e3 = 0, mag = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int k = 0; k < 16; ++k) ref += (double)A[i * 16 + k] * B[k * 32 + j];
      e1 = fmax(e1, fabs(D[i * 32 + j] - ref));
      e3 = fmax(e3, fabs(D3[i * 32 + j] - ref));
      mag = fmax(mag, fabs(ref));
    }
  printf("max|ref|=%.3f  bf16 (1 mfma) max err=%.3e   bf16x3 max err=%.3e\n", mag, e1, e3);
  printf(e1 < 0.05 && e3 < 1e-4 ? "LAYOUT_OK\n" : "LAYOUT_MISMATCH\n");
  return 0;
}
