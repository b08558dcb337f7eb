// Back-to-back MFMA issue-rate probe for gfx950: what the matrix pipe delivers with nothing else going on.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void bf16_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 7); y[e] = (__bf16)1.0f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void f32_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = (float)(threadIdx.x & 7), y = 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0];
  if (s == 123.456f) out[0] = s;
}

template <typename K>
static void run(const char* name, K kern, int wgs_per_cu, double flops_per_mfma, int nacc, int iters = 20000) {
  float* d;
  hipMalloc(&d, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double total = (double)grid * 4 * iters * nacc * flops_per_mfma;
  const double cyc_per_mfma = ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * nacc);
  printf("%-28s %d waves/SIMD: %.3f ms  %.1f TFLOP/s   (%.1f cycles per MFMA per SIMD at a nominal 2.4 GHz)\n", name,
         wgs_per_cu, ms, total / (ms * 1e-3) / 1e12, cyc_per_mfma);
  hipFree(d);
}

int main() {
  run("bf16 32x32x16, 8 accumulators", bf16_loop<8>, 1, 32768.0, 8);
  run("bf16 32x32x16, 8 accumulators", bf16_loop<8>, 2, 32768.0, 8);
  run("bf16 32x32x16, 4 accumulators", bf16_loop<4>, 1, 32768.0, 4);
  run("bf16 32x32x16, 2 accumulators", bf16_loop<2>, 1, 32768.0, 2);
  run("bf16 32x32x16, 1 accumulator ", bf16_loop<1>, 1, 32768.0, 1);
  run("fp32 32x32x2,  8 accumulators", f32_loop<8>, 1, 4096.0, 8);
  run("fp32 32x32x2,  8 accumulators", f32_loop<8>, 2, 4096.0, 8);
  run("fp32 32x32x2,  4 accumulators", f32_loop<4>, 1, 4096.0, 4);
  run("fp32 32x32x2,  2 accumulators", f32_loop<2>, 1, 4096.0, 2);
  run("fp32 32x32x2,  2 accumulators", f32_loop<2>, 2, 4096.0, 2);
  run("fp32 32x32x2,  2 accumulators", f32_loop<2>, 4, 4096.0, 2);
  run("fp32 32x32x2,  1 accumulator ", f32_loop<1>, 1, 4096.0, 1);
  run("fp32 32x32x2,  1 accumulator ", f32_loop<1>, 2, 4096.0, 1);
  // sustained: ~50 ms and ~0.5 s of back-to-back MFMAs (does the clock hold?)
  run("fp32 32x32x2, 8 acc, 50 ms ", f32_loop<8>, 2, 4096.0, 8, 120000);
  run("fp32 32x32x2, 8 acc, 0.5 s ", f32_loop<8>, 2, 4096.0, 8, 1200000);
  run("bf16 32x32x16, 8 acc, 50 ms", bf16_loop<8>, 2, 32768.0, 8, 200000);
  return 0;
}
