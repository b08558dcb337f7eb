// What do LDS reads cost an fp32 MFMA stream, and does it matter whether the accumulators are architectural (v) or accumulation
// (a) registers?  Two waves per SIMD, per iteration and wave 36 x v_mfma_f32_16x16x4_f32 (36 accumulators) in three runs of
// twelve; in front of every run NB ds_read_b128 + NS ds_read_b32 (conflict-free, lane-contiguous), one wait for all of them at
// the end of the iteration.  Round-3's mix probe had seen 20 ds_read_b128 per 64 MFMAs cost ~19 % (accumulators in VGPRs).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_return_under_mfma_probe.hip -o /tmp/ldsret && /tmp/ldsret
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int AGPR, int NB, int NS>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ a, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a[i & 2047];
  __syncthreads();
  f32x4 acc[36];
  for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  float wa[4], xb[4];
  for (int i = 0; i < 4; ++i) { wa[i] = a[(threadIdx.x + 64 * i) & 2047]; xb[i] = a[(threadIdx.x * 3 + 7 * i) & 2047]; }
  const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  const unsigned base4 = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096;
  f32x4 rb[NB > 0 ? NB : 1];
  float rs[NS > 0 ? NS : 1];
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int i = 0; i < NB; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[i]) : "v"(base), "n"(1024 * (i % 3)));
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rs[i]) : "v"(base4), "n"(256 * (i % 12)));
#pragma unroll
      for (int m = 12 * g; m < 12 * g + 12; ++m) {
        if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(wa[m & 3]), "v"(xb[(m >> 2) & 3]));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(wa[m & 3]), "v"(xb[(m >> 2) & 3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NB; ++i) asm volatile("" :: "v"(rb[i]));
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("" :: "v"(rs[i]));
    }
  }
  float s = sink;
  for (int m = 0; m < 36; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
  if (s == 123.456f) out[0] = s;
}

template <typename K>
static void run(const char* name, K kern, int iters = 4000) {
  const int grid = 512;
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *a, *o;
  (void)hipMalloc(&a, h.size() * 4); (void)hipMalloc(&o, 64);
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
  const double tf = (double)grid * 4 * iters * 36.0 * 2048 / (ms * 1e-3) / 1e12;
  printf("%-62s %.3f ms  %6.1f TFLOP/s\n", name, ms, tf);
  (void)hipFree(a); (void)hipFree(o);
}

int main() {
  for (int rep = 0; rep < 3; ++rep) run("(warm-up)", loop<0, 0, 0>, 40000);
  run("acc in VGPRs, no LDS reads", loop<0, 0, 0>);
  run("acc in AGPRs, no LDS reads", loop<1, 0, 0>);
  run("acc in VGPRs, 3 ds_read_b128 per 12 MFMAs", loop<0, 3, 0>);
  run("acc in AGPRs, 3 ds_read_b128 per 12 MFMAs", loop<1, 3, 0>);
  run("acc in VGPRs, 6 ds_read_b128 per 12 MFMAs", loop<0, 6, 0>);
  run("acc in AGPRs, 6 ds_read_b128 per 12 MFMAs", loop<1, 6, 0>);
  run("acc in VGPRs, 10 ds_read_b32 per 12 MFMAs", loop<0, 0, 10>);
  run("acc in AGPRs, 10 ds_read_b32 per 12 MFMAs", loop<1, 0, 10>);
  run("acc in VGPRs, 3 ds_read_b128 + 10 ds_read_b32 per 12 MFMAs", loop<0, 3, 10>);
  run("acc in AGPRs, 3 ds_read_b128 + 10 ds_read_b32 per 12 MFMAs", loop<1, 3, 10>);
  run("acc in VGPRs, no LDS reads (again)", loop<0, 0, 0>);
  return 0;
}
