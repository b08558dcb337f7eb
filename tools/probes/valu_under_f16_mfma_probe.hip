// Does VALU work overlap with a reduced-precision MFMA stream (it does not with fp32 MFMAs: pk_f32_under_mfma_probe.hip)?
// Two waves per SIMD, per iteration and wave 16 x v_mfma_f32_32x32x16_f16 (8 accumulators of 16 registers, two rounds) with NV
// VALU instructions (v_fma_f32 or 64-bit v_lshl_add_u64, as the InnerProduct kernels' copy addresses use) behind every MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_under_f16_mfma_probe.hip -o /tmp/vf16 && /tmp/vf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ a, float* out, int iters) {
  f32x16 acc[8];
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  f16x8 wa[2], xb[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 8; ++e) { wa[i][e] = (_Float16)a[(threadIdx.x + 64 * i + e) & 2047]; xb[i][e] = (_Float16)a[(threadIdx.x * 3 + 7 * i + e) & 2047]; }
  float p[8], q[8];
  unsigned long long ad[8];
  for (int i = 0; i < 8; ++i) { p[i] = a[(threadIdx.x + i) & 2047]; q[i] = a[(threadIdx.x + 5 * i) & 2047]; ad[i] = (unsigned long long)(a + i + threadIdx.x); }
  const float half = 0.5f;
  unsigned long long step = 128;
  asm volatile("" : "+s"(step));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(wa[m & 1]), "v"(xb[(m >> 1) & 1]));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int r = (m * NV + v) & 7;
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(half), "v"(q[r]));
        if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(ad[r]) : "s"(step));
      }
    }
  }
  float s = 0.f;
  for (int m = 0; m < 8; ++m)
    for (int e = 0; e < 16; ++e) s += acc[m][e];
  for (int i = 0; i < 8; ++i) s += p[i] + (float)(ad[i] & 3);
  if (s == 123.456f) out[0] = s;
}

template <typename K>
static void run(const char* name, K kern, int iters = 4000) {
  const int grid = 512;
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *a, *o;
  (void)hipMalloc(&a, h.size() * 4 + 65536); (void)hipMalloc(&o, 64);
  (void)hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, o, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, o, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)grid * 4 * iters * 16.0 * 32768 / (ms * 1e-3) / 1e12;
  printf("%-44s %.3f ms  %7.1f TFLOP/s\n", name, ms, tf);
  (void)hipFree(a); (void)hipFree(o);
}

int main() {
  for (int rep = 0; rep < 3; ++rep) run("(warm-up)", loop<0, 0>, 20000);
  run("f16 MFMA only", loop<0, 0>);
  run("+ 1 v_fma_f32 per MFMA", loop<1, 1>);
  run("+ 2 v_fma_f32 per MFMA", loop<1, 2>);
  run("+ 4 v_fma_f32 per MFMA", loop<1, 4>);
  run("+ 8 v_fma_f32 per MFMA", loop<1, 8>);
  run("+ 1 v_lshl_add_u64 per MFMA", loop<2, 1>);
  run("+ 2 v_lshl_add_u64 per MFMA", loop<2, 2>);
  run("f16 MFMA only (again)", loop<0, 0>);
  return 0;
}
