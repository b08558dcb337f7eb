// TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.
//
// A tiny "CUDA on the host CPU" execution shim.  oracle/build_ref.py force-includes
// this header when it compiles the reference's two CUDA translation units
//   /root/reference/lib/nms/nms_kernel.cu   and   /root/reference/lib/nms/mv_kernel.cu
// *where they lie* (no copy of the sources is made) with plain g++, producing
// oracle/_ref/libmnc_ref.so whose exported `_nms` / `_mv` are the reference's own code.
//
// What the shim provides:
//   * __global__/__device__/__shared__ as no-ops / `static` (one block runs at a time per OS thread)
//   * blockIdx/threadIdx/blockDim/gridDim as thread_local variables
//   * cudaMalloc/cudaMemcpy/cudaFree/... mapped onto malloc/memcpy/free
//   * a kernel launcher (the build script rewrites `k<<<g,b>>>(args)` into MNC_CPU_LAUNCH(...)):
//       - "direct" mode: every (block, thread) executes the kernel body to completion in turn,
//         blocks distributed over OpenMP threads (legal for kernels without __syncthreads)
//       - "fiber" mode: the threads of one block are ucontext fibers; __syncthreads() yields to the
//         next fiber, so barrier semantics are exact (used for nms_kernel)
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct mnc_uint3 { unsigned x, y, z; };

extern thread_local mnc_uint3 blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

// CUDA's overloaded device math used unqualified by the reference kernels
static inline float max(float a, float b) { return a > b ? a : b; }
static inline float min(float a, float b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
using std::floor;

// ---- runtime API subset -------------------------------------------------------------------------
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
static inline const char* cudaGetErrorString(cudaError_t) { return "cuda-on-cpu error"; }
template <typename T>
static inline cudaError_t cudaMalloc(T** p, size_t bytes) {
  *p = (T*)malloc(bytes ? bytes : 1);
  return *p ? cudaSuccess : 1;
}
static inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) {
  memcpy(dst, src, n);
  return cudaSuccess;
}
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
static inline cudaError_t cudaPeekAtLastError() { return cudaSuccess; }

// ---- launcher -----------------------------------------------------------------------------------
void mnc_cpu_launch_direct(dim3 grid, dim3 block, const std::function<void()>& body);
void mnc_cpu_launch_fiber(dim3 grid, dim3 block, const std::function<void()>& body);
void mnc_cpu_syncthreads();
#define __syncthreads() mnc_cpu_syncthreads()

#define MNC_CPU_LAUNCH(mode, kernel, grid, block, ...) \
  mnc_cpu_launch_##mode(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
