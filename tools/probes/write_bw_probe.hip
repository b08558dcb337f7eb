// How fast can this part WRITE?  conv1_1 produces 153.6 MB from 7.2 MB of input; two unrelated kernels (a VALU kernel with 2 KB
// stores per wave instruction pair, an MFMA kernel with 512-byte runs) both take 60 us = 2.6 TB/s.  This probe times pure stores of
// the same 153.6 MB (float4 per lane, grid-stride; 32 bytes per lane; nontemporal; hipMemsetAsync) next to a copy of the same size.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/write_bw_probe.hip -o /tmp/write_bw_probe && /tmp/write_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void fill16(float4* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void fill32(float4* p, long n, float v) {        // 32 contiguous bytes per lane (one c8 pixel)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += (long)gridDim.x * blockDim.x) {
    p[2 * i] = make_float4(v, v, v, v);
    p[2 * i + 1] = make_float4(v, v, v, v);
  }
}
__global__ void fill16_nt(float4* p, long n, float v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(f4{v, v, v, v}, reinterpret_cast<f4*>(p) + i);
}
__global__ void copy16(const float4* __restrict__ s, float4* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}

template <class F>
static double time_us(F f, int reps = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / reps;
}

int main() {
  const long bytes = 64L * 600 * 1000 * 4;                // conv1_1's output
  const long n = bytes / 16;
  float4 *d = nullptr, *s = nullptr;
  hipMalloc(&d, bytes); hipMalloc(&s, bytes);
  hipMemset(s, 0, bytes);
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    const double t16 = time_us([&] { hipLaunchKernelGGL(fill16, dim3(grid), dim3(256), 0, 0, d, n, 1.f); });
    const double t32 = time_us([&] { hipLaunchKernelGGL(fill32, dim3(grid), dim3(256), 0, 0, d, n, 1.f); });
    const double tnt = time_us([&] { hipLaunchKernelGGL(fill16_nt, dim3(grid), dim3(256), 0, 0, d, n, 1.f); });
    const double tcp = time_us([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, s, d, n); });
    printf("grid %5d: fill 16 B/lane %6.1f us = %5.2f TB/s | 32 B/lane %6.1f us = %5.2f TB/s | nontemporal %6.1f us = %5.2f TB/s | copy %6.1f us = %5.2f TB/s (read + write)\n",
           grid, t16, bytes / t16 / 1e6, t32, bytes / t32 / 1e6, tnt, bytes / tnt / 1e6, tcp, 2.0 * bytes / tcp / 1e6);
  }
  const double tm = time_us([&] { hipMemsetAsync(d, 0, bytes, 0); });
  printf("hipMemsetAsync: %6.1f us = %5.2f TB/s\n", tm, bytes / tm / 1e6);
  return 0;
}
