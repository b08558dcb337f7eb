// Does the fp32 MFMA stream lose time to LDS reads and VALU results landing in the register file, and does the tile shape
// matter?  Per iteration and wave: NREAD ds_read_b128 (conflict-free), NVALU v_add_f32 on the loaded values, and the same
// FLOPs either as 32 x v_mfma_f32_32x32x2_f32 (16 result registers per 4096 flop) or 64 x v_mfma_f32_16x16x4_f32 (4 per 2048).
// Reports shader cycles per iteration and wave (s_memtime) and the clock (wall_clock64), two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_mix_probe.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SMALL, int NREAD, int NVALU>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ a, float* out, unsigned long long* stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 24 * 1024 / 4 * 4; i += 256) lds[i] = a[i & 2047];
  __syncthreads();
  f32x16 acc[8];
  f32x4 acs[32];
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  for (int p = 0; p < 32; ++p) acs[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* base = lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 64;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x4 r[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) r[i] = *reinterpret_cast<const f32x4*>(base + (i % 20) * 256);
  for (int it = 0; it < iters; ++it) {
    float op[64];
#pragma unroll
    for (int i = 0; i < 16; ++i) { op[4 * i] = r[i].x; op[4 * i + 1] = r[i].y; op[4 * i + 2] = r[i].z; op[4 * i + 3] = r[i].w; }
#pragma unroll
    for (int i = 0; i < NVALU; ++i) op[i % 32] = op[i % 32] - op[32 + (i + 7) % 32];
    const int off = (it & 1) * 64;
#pragma unroll
    for (int i = 0; i < NREAD; ++i) r[i] = *reinterpret_cast<const f32x4*>(base + off + i * 256);
    if (SMALL) {
#pragma unroll
      for (int p = 0; p < 32; ++p) {
        acs[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[32 + p % 32], op[p % 32], acs[p], 0, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < 32; ++p) {
        acs[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[32 + (p + 5) % 32], op[(p + 3) % 32], acs[p], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; p += 2)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[32 + p * 4 + k], op[p * 4 + k], acc[p], 0, 0, 0);
          acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[32 + p * 4 + 4 + k], op[p * 4 + 4 + k], acc[p + 1], 0, 0, 0);
        }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int p = 0; p < 8; ++p)
    for (int e = 0; e < 16; ++e) s += acc[p][e];
  for (int p = 0; p < 32; ++p) s += acs[p].x + acs[p].y + acs[p].z + acs[p].w;
  for (int i = 0; i < 20; ++i) s += r[i].x;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = w1 - w0;
  }
}

template <typename K>
static void run(const char* name, K kern, int iters = 3000) {
  const int grid = 512;
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *a, *o;
  unsigned long long* st;
  hipMalloc(&a, h.size() * 4); hipMalloc(&o, 64); hipMalloc(&st, grid * 4 * 16);
  hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 64 * 1024, 0, a, o, st, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 64 * 1024, 0, a, o, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(grid * 4 * 2);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < grid * 4; ++w) { cyc += s[2 * w]; wall += s[2 * w + 1]; }
  cyc /= grid * 4; wall /= grid * 4;
  const double tf = (double)grid * 4 * iters * 32.0 * 4096 / (ms * 1e-3) / 1e12;
  printf("%-44s %.3f ms  %6.1f TFLOP/s | %.0f shader cycles per iteration and wave (2 waves/SIMD; 32 MFMAs of 64 = 2048), clock %.2f GHz\n",
         name, ms, tf, cyc / iters, cyc / (wall * 10.0));
  hipFree(a); hipFree(o); hipFree(st);
}

int main() {
  run("32x32x2  bare", loop<0, 0, 0>);
  run("16x16x4  bare", loop<1, 0, 0>);
  run("32x32x2  + 20 ds_read_b128", loop<0, 20, 0>);
  run("16x16x4  + 20 ds_read_b128", loop<1, 20, 0>);
  run("32x32x2  + 64 v_sub", loop<0, 0, 64>);
  run("16x16x4  + 64 v_sub", loop<1, 0, 64>);
  run("32x32x2  + 20 ds_read_b128 + 64 v_sub", loop<0, 20, 64>);
  run("16x16x4  + 20 ds_read_b128 + 64 v_sub", loop<1, 20, 64>);
  return 0;
}
