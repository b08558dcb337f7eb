// What does a v_mfma_f32_32x32x16_f16 stream sustain on gfx950 -- shader cycles per instruction AND the clock the part holds under
// it -- with constant and with random operands?  (The companion of mfma_f32_clock_probe.hip: 2.5 PFLOP/s = 2.4 GHz x 32 cycles per
// instruction is the nameplate the reduced-precision kernels are priced against in bench.py.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f16_clock_probe.hip -o /tmp/f16clk && /tmp/f16clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void loop(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float* out,
                                               unsigned long long* stamps, int iters) {
  f32x16 acc[8];
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  f16x8 x[4], y[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { x[i][e] = a[(threadIdx.x * 4 + i) * 8 + e]; y[i][e] = b[(threadIdx.x * 4 + i) * 8 + e]; }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[k], y[(k + p) & 3], acc[p], 0, 0, 0);
    asm volatile("" : "+v"(x[0]), "+v"(y[0]));
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) s += acc[p][e];
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = w1 - w0;
  }
}

static void run(int wgs_per_cu, int mode, int iters = 8000) {
  const int grid = 256 * wgs_per_cu;
  std::vector<_Float16> h(256 * 4 * 8);
  for (auto& v : h) v = (_Float16)(mode == 0 ? 0.f : mode == 1 ? 1.f : (float)rand() / RAND_MAX * 2.f - 1.f);
  _Float16 *a, *b;
  float* o;
  unsigned long long* st;
  hipMalloc(&a, h.size() * 2); hipMalloc(&b, h.size() * 2); hipMalloc(&o, 64); hipMalloc(&st, grid * 4 * 16);
  hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(loop, dim3(grid), dim3(256), 0, 0, a, b, o, st, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(loop, dim3(grid), dim3(256), 0, 0, a, b, o, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(grid * 4 * 2);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < grid * 4; ++w) { cyc += s[2 * w]; wall += s[2 * w + 1]; }
  const double mfmas = 32.0 * iters;
  const double tf = (double)grid * 4 * mfmas * 32768 / (ms * 1e-3) / 1e12;
  printf("32x32x16 f16, %d wave(s)/SIMD, operands %-8s: %.3f ms  %7.1f TFLOP/s | per wave %.1f shader cycles per MFMA, shader clock %.2f GHz\n",
         wgs_per_cu, mode == 0 ? "zeros" : mode == 1 ? "ones" : "random", ms, tf, cyc / (grid * 4) / mfmas, cyc / (wall * 10.0));
  hipFree(a); hipFree(b); hipFree(o); hipFree(st);
}

int main() {
  for (int mode = 0; mode < 3; ++mode)
    for (int w = 1; w <= 2; ++w) run(w, mode);
  run(2, 2, 80000);
  return 0;
}
