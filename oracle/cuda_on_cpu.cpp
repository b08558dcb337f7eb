// TEST INFRASTRUCTURE ONLY -- runtime half of oracle/cuda_on_cpu.h (see that header).
#include "cuda_on_cpu.h"

thread_local mnc_uint3 blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;

void mnc_cpu_launch_direct(dim3 grid, dim3 block, const std::function<void()>& body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 64)
  for (long b = 0; b < nblocks; ++b) {
    gridDim = grid;
    blockDim = block;
    blockIdx.x = (unsigned)(b % grid.x);
    blockIdx.y = (unsigned)((b / grid.x) % grid.y);
    blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
    for (unsigned tz = 0; tz < block.z; ++tz)
      for (unsigned ty = 0; ty < block.y; ++ty)
        for (unsigned tx = 0; tx < block.x; ++tx) {
          threadIdx.x = tx; threadIdx.y = ty; threadIdx.z = tz;
          body();
        }
  }
}

// ---- fiber mode: exact __syncthreads() ----------------------------------------------------------
namespace {
struct FiberBlock {
  std::vector<ucontext_t> ctx;
  std::vector<char> done;
  std::vector<std::vector<char>> stacks;
  ucontext_t sched;
  int cur = -1;
  const std::function<void()>* body = nullptr;
};
thread_local FiberBlock* g_fb = nullptr;

void fiber_entry() {
  FiberBlock* fb = g_fb;
  (*fb->body)();
  fb->done[fb->cur] = 1;
  swapcontext(&fb->ctx[fb->cur], &fb->sched);
}
}  // namespace

void mnc_cpu_syncthreads() {
  FiberBlock* fb = g_fb;
  if (!fb) {
    fprintf(stderr, "cuda_on_cpu: __syncthreads() in a kernel launched in direct mode\n");
    abort();
  }
  // Round-robin: yielding to the scheduler lets every other fiber reach its barrier before we resume.
  swapcontext(&fb->ctx[fb->cur], &fb->sched);
}

void mnc_cpu_launch_fiber(dim3 grid, dim3 block, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  const size_t kStack = 64 * 1024;
  FiberBlock fb;
  fb.ctx.resize(nthreads);
  fb.done.resize(nthreads);
  fb.stacks.assign(nthreads, std::vector<char>(kStack));
  fb.body = &body;
  g_fb = &fb;
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
        for (int t = 0; t < nthreads; ++t) {
          getcontext(&fb.ctx[t]);
          fb.ctx[t].uc_stack.ss_sp = fb.stacks[t].data();
          fb.ctx[t].uc_stack.ss_size = kStack;
          fb.ctx[t].uc_link = &fb.sched;
          makecontext(&fb.ctx[t], fiber_entry, 0);
          fb.done[t] = 0;
        }
        int remaining = nthreads;
        while (remaining > 0) {
          // one sweep == one barrier phase: every live fiber runs until its next __syncthreads()/exit
          for (int t = 0; t < nthreads; ++t) {
            if (fb.done[t]) continue;
            fb.cur = t;
            threadIdx.x = t % block.x;
            threadIdx.y = (t / block.x) % block.y;
            threadIdx.z = t / (block.x * block.y);
            swapcontext(&fb.sched, &fb.ctx[t]);
            if (fb.done[t]) --remaining;
          }
        }
      }
  g_fb = nullptr;
}
