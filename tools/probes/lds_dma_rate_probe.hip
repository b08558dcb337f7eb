// What does a CU's global -> LDS path deliver when every CU streams operand panels by LDS-DMA the way fc_lowp_dma_kernel does (eight
// waves, 1 KB per wave instruction, one barrier per stage) -- with no MFMA at all?  The reduced-precision InnerProducts are priced
// against this rate.
//   * pure streams: `span` bytes walked cyclically by every workgroup from its own offset (small span: L2 / Infinity-Cache resident;
//     large: HBM), 72 KB stages, ONE stage in flight (issue, wait, barrier);
//   * the InnerProduct's mix at 300 RoIs: per stage 5/9 of the bytes from a cache-resident region (the activation panel every column
//     tile re-reads) and 4/9 from an HBM-sized one (the weights, read once), with DEPTH stages in flight in a ring of DEPTH + 1 buffers:
//     is a stage's round trip latency bound, i.e. does a deeper ring help?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_rate_probe.hip -o /tmp/dma && /tmp/dma
#include <hip/hip_runtime.h>
#include <cstdio>

template <int STAGE_KB, int DEPTH, int MIX>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, size_t span, const char* __restrict__ hot, int stages,
                                              float* out) {
  constexpr int kStage = STAGE_KB * 1024, kPer = STAGE_KB / 8;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  size_t off = ((size_t)blockIdx.x * 7919 * kStage) % span;
  size_t hoff = ((size_t)(blockIdx.x & 15) * 577 * 1024) % ((size_t)8 << 20);
  auto issue = [&](int s) {
    char* dst = lds + (s % (DEPTH + 1)) * kStage;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int piece = wave + 8 * i;
      const char* g = (MIX && piece * 9 < kPer * 8 * 5) ? hot + hoff + piece * 1024 : src + off + piece * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
    off += kStage;
    if (off + kStage > span) off = 0;
    hoff += kStage;
    if (hoff + kStage > ((size_t)8 << 20)) hoff = 0;
  };
  for (int s = 0; s < DEPTH - 1; ++s) issue(s);
  for (int s = 0; s < stages; ++s) {
    issue(s + DEPTH - 1);
    // stage s has landed when at most (DEPTH - 1) stages' pieces are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * kPer) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && lds[blockIdx.x & 1023] == 123) out[0] = 1.f;
}

template <typename K>
static void run(const char* what, K kern, int stage_kb, int depth, const char* src, size_t span, const char* hot, float* out) {
  const int lds = (depth + 1) * stage_kb * 1024, stages = 28800 / stage_kb, grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, src, span, hot, 20, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, src, span, hot, stages, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * stages * stage_kb * 1024;
  printf("%-34s span %5zu MB, %2d KB stages, %d in flight: %.3f ms  %5.2f TB/s = %4.1f B/clk/CU (nominal 2.4 GHz), %.2f us per 72 KB\n", what,
         span >> 20, stage_kb, depth, ms, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9,
         ms * 1e3 / stages * 72.0 / stage_kb);
}

int main() {
  const size_t big = (size_t)2 << 30;
  char *src, *hot;
  float* out;
  hipMalloc(&src, big);
  hipMemset(src, 1, big);
  hipMalloc(&hot, (size_t)9 << 20);
  hipMemset(hot, 1, (size_t)9 << 20);
  hipMalloc(&out, 64);
  for (size_t span : {(size_t)8 << 20, (size_t)192 << 20, big}) run("pure stream", stream<72, 1, 0>, 72, 1, src, span, hot, out);
  run("pure stream", stream<36, 3, 0>, 36, 3, src, big, hot, out);
  run("mix 5/9 cached + 4/9 HBM", stream<72, 1, 1>, 72, 1, src, big, hot, out);
  run("mix 5/9 cached + 4/9 HBM", stream<36, 1, 1>, 36, 1, src, big, hot, out);
  run("mix 5/9 cached + 4/9 HBM", stream<36, 2, 1>, 36, 2, src, big, hot, out);
  run("mix 5/9 cached + 4/9 HBM", stream<36, 3, 1>, 36, 3, src, big, hot, out);
  run("mix 5/9 cached + 4/9 HBM", stream<24, 5, 1>, 24, 5, src, big, hot, out);
  return 0;
}
