// What does a v_mfma_f32_32x32x2_f32 stream cost on gfx950, in SHADER cycles and in wall time?  Separates the pipe's issue rate
// from the clock the part sustains under that load (DVFS): every wave times its loop with s_memtime (shader cycles) and
// wall_clock64 (100 MHz).  Variants: operand values (zeros / random), accumulators per wave (8: the Winograd wave-pair loop),
// dependent distance (alternating pairs as in mfma_pair, or round robin), one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_clock_probe.hip -o /tmp/mfma_clk && /tmp/mfma_clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PAIRS>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ a, const float* __restrict__ b, float* out,
                                               unsigned long long* stamps, int iters) {
  f32x16 acc[8];
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = a[threadIdx.x * 8 + i]; y[i] = b[threadIdx.x * 8 + i]; }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (PAIRS) {
#pragma unroll
      for (int p = 0; p < 8; p += 2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k], y[p], acc[p], 0, 0, 0);
          acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k + 4], y[p + 1], acc[p + 1], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k], y[p], acc[p], 0, 0, 0);
    }
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

template <typename K>
static void run(const char* name, K kern, int wgs_per_cu, bool random, int iters = 4000) {
  const int grid = 256 * wgs_per_cu;
  std::vector<float> h(256 * 8);
  for (auto& v : h) v = random ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
  float *a, *b, *o;
  unsigned long long* st;
  hipMalloc(&a, h.size() * 4); hipMalloc(&b, h.size() * 4); hipMalloc(&o, 64); hipMalloc(&st, grid * 4 * 16);
  hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, o, st, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, o, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(grid * 4 * 2);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < grid * 4; ++w) { cyc += s[2 * w]; wall += s[2 * w + 1]; }
  cyc /= grid * 4; wall /= grid * 4;
  const double mfmas = 32.0 * iters;                       // per wave
  const double tf = (double)grid * 4 * mfmas * 4096 / (ms * 1e-3) / 1e12;
  printf("%-22s %d wave(s)/SIMD %-6s: %.3f ms  %.1f TFLOP/s | per wave: %.1f shader cycles per MFMA, shader clock %.2f GHz\n", name,
         wgs_per_cu, random ? "random" : "zeros", ms, tf, cyc / mfmas, cyc / (wall * 10.0));
  hipFree(a); hipFree(b); hipFree(o); hipFree(st);
}

int main() {
  for (int rnd = 0; rnd < 2; ++rnd)
    for (int w = 1; w <= 2; ++w) {
      run("alternating pairs", loop<1>, w, rnd);
      run("round robin (8 accs)", loop<0>, w, rnd);
    }
  // longer run: does the clock hold?
  run("alternating pairs", loop<1>, 2, true, 40000);
  return 0;
}
