// Can packed fp32 VALU (v_pk_fma_f32) hide under an fp32 MFMA stream the way scalar v_fma_f32 does?  (hipcc's pre-emit peephole
// UNPACKS v_pk_*_f32 it finds in the shadow of an MFMA, which suggests it cannot.)  Two waves per SIMD, per iteration and wave 36 x
// v_mfma_f32_16x16x4_f32 (36 accumulators, the F(4x4) convolution's column-by-column stream) with NV VALU instructions behind every
// MFMA: scalar v_fma_f32 or v_pk_fma_f32 (inline assembly: the compiler neither reorders nor unpacks them).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_under_mfma_probe.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: nothing; 1: NV x v_fma_f32; 2: NV x v_pk_fma_f32; 3: NV x v_pk_add_f32; 4: NV x v_pk_fma_f32 with op_sel broadcasts
template <int KIND, int NV, int G = 1, int D = 8>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ a, float* out, unsigned long long* stamps, int iters) {
  f32x4 acc[36];
  for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  float wa[4], xb[4];
  for (int i = 0; i < 4; ++i) { wa[i] = a[(threadIdx.x + 64 * i) & 2047]; xb[i] = a[(threadIdx.x * 3 + 7 * i) & 2047]; }
  f32x2 p[8], q[8];
  for (int i = 0; i < 8; ++i) {
    p[i] = f32x2{a[(threadIdx.x + i) & 2047], a[(threadIdx.x + 9 * i) & 2047]};
    q[i] = f32x2{a[(threadIdx.x + 5 * i) & 2047], a[(threadIdx.x + 11 * i) & 2047]};
  }
  const f32x2 half = {0.5f, 0.25f};
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m0 = 0; m0 < 36; m0 += G) {             // G MFMAs back to back, then their G * NV VALU instructions
#pragma unroll
      for (int m = m0; m < m0 + G; ++m)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(wa[m & 3]), "v"(xb[(m >> 2) & 3]));
#pragma unroll
      for (int v = 0; v < NV * G; ++v) {
        const int r = (m0 * NV + v) % D;          // D = dependency distance of the VALU chain
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[r].x) : "v"(half.x), "v"(q[r].x));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(half), "v"(q[r]));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[r]) : "v"(half), "v"(q[r]));
        if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "+v"(p[r]) : "v"(half), "v"(q[r]));
        if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(p[r].x) : "v"(q[r].x));
        if (KIND == 6) asm volatile("s_nop 3");
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int m = 0; m < 36; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  if (s == 123.456f) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = w1 - w0;
  }
}

template <typename K>
static void run(const char* name, K kern, int iters = 4000) {
  const int grid = 512;
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *a, *o;
  unsigned long long* st;
  hipMalloc(&a, h.size() * 4); hipMalloc(&o, 64); hipMalloc(&st, grid * 4 * 16);
  hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, o, st, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, o, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> s(grid * 4 * 2);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < grid * 4; ++w) { cyc += s[2 * w]; wall += s[2 * w + 1]; }
  cyc /= grid * 4; wall /= grid * 4;
  const double tf = (double)grid * 4 * iters * 36.0 * 2048 / (ms * 1e-3) / 1e12;
  printf("%-40s %.3f ms  %6.1f TFLOP/s | %.1f shader cycles per MFMA and wave (2 waves/SIMD: 64 = pipe full), clock %.2f GHz\n", name, ms,
         tf, cyc / iters / 36.0, cyc / (wall * 10.0));
  hipFree(a); hipFree(o); hipFree(st);
}

int main() {
  for (int rep = 0; rep < 3; ++rep) run("(warm-up) MFMA only", loop<0, 0>, 40000);
  for (int rep = 0; rep < 1; ++rep) {
  printf("-- pass %d\n", rep);
  run("MFMA only", loop<0, 0>);
  run("+ 2 v_fma_f32 per MFMA", loop<1, 2>);
  run("+ 4 v_fma_f32 per MFMA", loop<1, 4>);
  run("+ 1 v_pk_fma_f32 per MFMA", loop<2, 1>);
  run("+ 2 v_pk_fma_f32 per MFMA", loop<2, 2>);
  run("+ 4 v_pk_fma_f32 per MFMA", loop<2, 4>);
  run("+ 2 v_pk_fma_f32 (op_sel) per MFMA", loop<4, 2>);
  run("6 MFMA, then 12 v_pk_fma_f32", loop<2, 2, 6>);
  run("6 MFMA, then 24 v_fma_f32", loop<1, 4, 6>);
  run("36 MFMA, then 72 v_pk_fma_f32", loop<2, 2, 36>);
  run("+ 2 v_mov_b32 per MFMA", loop<5, 2>);
  run("+ 2 s_nop 3 per MFMA", loop<6, 2>);
  run("6 MFMA, then 12 v_pk_fma_f32, chain distance 1", loop<2, 2, 6, 1>);
  run("6 MFMA, then 12 v_pk_fma_f32, chain distance 2", loop<2, 2, 6, 2>);
  run("6 MFMA, then 12 v_pk_fma_f32, chain distance 3", loop<2, 2, 6, 3>);
  run("6 MFMA, then 24 v_fma_f32, chain distance 1", loop<1, 4, 6, 1>);
  run("6 MFMA, then 24 v_fma_f32, chain distance 2", loop<1, 4, 6, 2>);
  run("6 MFMA, then 6 v_pk_fma_f32", loop<2, 1, 6>);
  run("6 MFMA, then 18 v_pk_fma_f32", loop<2, 3, 6>);
  run("MFMA only (again)", loop<0, 0>);
  }
  return 0;
}
