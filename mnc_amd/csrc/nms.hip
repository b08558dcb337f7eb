// NMS for gfx950 -- replaces lib/nms/nms_kernel.cu (reference).  Compiled with -ffp-contract=off: the IoU is
// evaluated operation by operation exactly as devIoU (nms_kernel.cu:24-32) writes it, so the suppression bitmask and
// the keep list are bit-exact with the reference.
//
// Design (wave64-first, not a translation of the 64-thread CUDA block):
//   * one 64x64 tile of the suppression matrix == one wavefront == one 64-bit ballot.  Lane j owns COLUMN box j in
//     registers; the wave walks the 64 ROW boxes, whose coordinates are wave-uniform (LDS broadcast reads), and
//     __ballot(IoU > thr) IS row i's 64-bit mask word.  Lane i keeps word i, so the final store is one coalesced-per-row
//     8-byte store per lane.  Four waves (four column tiles) share one row tile through LDS.
//   * only the upper triangle of tiles is computed; the reference launches all col_blocks^2 tiles (nms_kernel.cu:39
//     has the early exit commented out) although its scan never reads the lower ones (:135).
//   * the greedy scan runs on the device in ONE wave: per 64-box block, the 64 diagonal words are fetched with one
//     vector load and the intra-block chain is resolved with v_readlane only; the rows of the survivors are then OR-ed
//     into the per-lane `remv` words with independent (non-chained) loads.  So the serial dependency is one load per
//     block of 64 boxes instead of one per kept box, and the 4.5 MB mask never crosses PCIe.
#include <mutex>

#include "mnc_internal.h"

namespace mnc {

typedef unsigned long long u64;

__device__ __forceinline__ float iou_ref_order(float a0, float a1, float a2, float a3, float Sa, float b0, float b1,
                                               float b2, float b3, float Sb) {
  // devIoU(a, b), nms_kernel.cu:24-32
  float left = fmaxf(a0, b0), right = fminf(a2, b2);
  float top = fmaxf(a1, b1), bottom = fminf(a3, b3);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  return interS / (Sa + Sb - interS);
}

constexpr int kWavesPerBlock = 4;

// grid: (ceil(cb / 4), cb); block: 256.  boxes: [n][dim] sorted by descending score.  mask: [n][cb].
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, int n, int dim, float thr,
                                                       u64* __restrict__ mask, int cb) {
  const int rt = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ct = blockIdx.x * kWavesPerBlock + wave;
  __shared__ float rowbox[64][4];
  if (blockIdx.x * kWavesPerBlock + kWavesPerBlock - 1 < rt) return;  // whole block below the diagonal
  if (threadIdx.x < 64) {
    const int r = rt * 64 + threadIdx.x;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) v = make_float4(boxes[(long)r * dim + 0], boxes[(long)r * dim + 1], boxes[(long)r * dim + 2],
                               boxes[(long)r * dim + 3]);
    rowbox[threadIdx.x][0] = v.x; rowbox[threadIdx.x][1] = v.y; rowbox[threadIdx.x][2] = v.z; rowbox[threadIdx.x][3] = v.w;
  }
  __syncthreads();
  if (ct >= cb || ct < rt) return;

  const int c = ct * 64 + lane;
  const bool cvalid = c < n;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  if (cvalid) {
    b0 = boxes[(long)c * dim + 0]; b1 = boxes[(long)c * dim + 1];
    b2 = boxes[(long)c * dim + 2]; b3 = boxes[(long)c * dim + 3];
  }
  const float Sb = (b2 - b0 + 1) * (b3 - b1 + 1);
  const bool diag = (ct == rt);
  const int rows = min(n - rt * 64, 64);
  u64 mine = 0;
  for (int i = 0; i < rows; ++i) {
    const float a0 = rowbox[i][0], a1 = rowbox[i][1], a2 = rowbox[i][2], a3 = rowbox[i][3];
    const float Sa = (a2 - a0 + 1) * (a3 - a1 + 1);
    const float ov = iou_ref_order(a0, a1, a2, a3, Sa, b0, b1, b2, b3, Sb);
    const bool hit = cvalid && (!diag || lane > i) && (ov > thr);  // strict >, nms_kernel.cu:71; j > i on the diagonal, :66-69
    const u64 word = __ballot(hit);
    if (lane == i) mine = word;
  }
  if (lane < rows) mask[(long)(rt * 64 + lane) * cb + ct] = mine;
}

constexpr int kMaxWordsPerLane = 8;  // device scan handles cb <= 512, i.e. n <= 32768

// One wave.  keep: capacity n.  Greedy scan of nms_kernel.cu:124-140, stopping after max_keep survivors.
__global__ __launch_bounds__(64) void nms_scan_kernel(const u64* __restrict__ mask, int n, int cb, int max_keep,
                                                      int* __restrict__ keep, int* __restrict__ num_out) {
  const int lane = threadIdx.x;
  u64 remv[kMaxWordsPerLane];
#pragma unroll
  for (int q = 0; q < kMaxWordsPerLane; ++q) remv[q] = 0;
  int nk = 0;
  for (int b = 0; b < cb && nk < max_keep; ++b) {
    // current removal word of block b lives in lane (b & 63), slot (b >> 6)
    u64 slot = 0;
#pragma unroll
    for (int q = 0; q < kMaxWordsPerLane; ++q)
      if (q == (b >> 6)) slot = remv[q];
    u64 cur = __shfl(slot, b & 63);
    const int row = b * 64 + lane;
    const u64 dword = row < n ? mask[(long)row * cb + b] : 0ull;
    const int rows = min(n - b * 64, 64);
    u64 keptbits = 0;
    for (int i = 0; i < rows && nk < max_keep; ++i) {
      const u64 di = __shfl(dword, i);
      if (!((cur >> i) & 1ull)) {
        keptbits |= 1ull << i;
        if (lane == 0) keep[nk] = b * 64 + i;
        ++nk;
        cur |= di;
      }
    }
    // fold the survivors' rows into remv for the words this lane owns (independent loads)
    u64 bits = keptbits;
    while (bits) {
      const int i = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      const long base = (long)(b * 64 + i) * cb;
#pragma unroll
      for (int q = 0; q < kMaxWordsPerLane; ++q) {
        const int w = lane + 64 * q;
        if (w > b && w < cb) remv[q] |= mask[base + w];
      }
    }
  }
  if (lane == 0) *num_out = nk;
}

// ---- launchers ------------------------------------------------------------------------------------------------
int nms_mask_launch(hipStream_t stream, const float* d_boxes, int n, int dim, float thr, u64* d_mask) {
  const int cb = cdiv(n, 64);
  dim3 grid(cdiv(cb, kWavesPerBlock), cb);
  hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, stream, d_boxes, n, dim, thr, d_mask, cb);
  return MNC_OK;
}

int nms_scan_launch(hipStream_t stream, const u64* d_mask, int n, int max_keep, int* d_keep, int* d_num) {
  const int cb = cdiv(n, 64);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, d_mask, n, cb, max_keep, d_keep, d_num);
  return MNC_OK;
}

// ---- per-device workspace for the host-pointer entry points (b1/b2 allocate-per-call in the reference) ----------
static LegacyWs g_ws[16];

int legacy_ws(int device_id, size_t bytes, LegacyWs** out) {
  int ndev = 0;
  MNC_HIP_TRY(hipGetDeviceCount(&ndev));
  MNC_REQUIRE(device_id >= 0 && device_id < ndev && device_id < 16, "device %d out of range (have %d)", device_id, ndev);
  MNC_HIP_TRY(hipSetDevice(device_id));
  LegacyWs* w = &g_ws[device_id];
  if (!w->stream) MNC_HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  if (bytes > w->cap) {
    if (w->buf) MNC_HIP_TRY(hipFree(w->buf));
    w->buf = nullptr;
    w->cap = 0;
    size_t want = bytes + (bytes >> 1) + 4096;
    hipError_t e = hipMalloc(&w->buf, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
      return MNC_ERR_NOMEM;
    }
    w->cap = want;
  }
  *out = w;
  return MNC_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int nms_host_impl(int* keep_out, int* num_out, u64* mask_out, const float* boxes_host, int n, int dim, float thr,
                         int max_keep, int device_id) {
  MNC_REQUIRE(n >= 0 && dim >= 4, "mnc_nms: boxes_num=%d boxes_dim=%d", n, dim);
  if (num_out) *num_out = 0;
  if (n == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(boxes_host, "mnc_nms: null boxes");
  const int cb = cdiv(n, 64);
  if (max_keep < 0 || max_keep > n) max_keep = n;
  const size_t box_b = align256((size_t)n * dim * 4), mask_b = align256((size_t)n * cb * 8), keep_b = align256((size_t)n * 4);
  LegacyWs* w = nullptr;
  int rc = legacy_ws(device_id, box_b + mask_b + keep_b + 256, &w);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(w->mu);
  char* base = (char*)w->buf;
  float* d_boxes = (float*)base;
  u64* d_mask = (u64*)(base + box_b);
  int* d_keep = (int*)(base + box_b + mask_b);
  int* d_num = (int*)(base + box_b + mask_b + keep_b);
  MNC_HIP_TRY(hipMemcpyAsync(d_boxes, boxes_host, (size_t)n * dim * 4, hipMemcpyHostToDevice, w->stream));
  const bool device_scan = cb <= 64 * kMaxWordsPerLane && !mask_out;
  if (mask_out) MNC_HIP_TRY(hipMemsetAsync(d_mask, 0, (size_t)n * cb * 8, w->stream));
  nms_mask_launch(w->stream, d_boxes, n, dim, thr, d_mask);
  MNC_HIP_TRY(hipGetLastError());
  if (mask_out) {
    MNC_HIP_TRY(hipMemcpyAsync(mask_out, d_mask, (size_t)n * cb * 8, hipMemcpyDeviceToHost, w->stream));
    MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    clear_error();
    return MNC_OK;
  }
  if (device_scan) {
    nms_scan_launch(w->stream, d_mask, n, max_keep, d_keep, d_num);
    MNC_HIP_TRY(hipGetLastError());
    int nk = 0;
    MNC_HIP_TRY(hipMemcpyAsync(&nk, d_num, 4, hipMemcpyDeviceToHost, w->stream));
    MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    if (nk > 0) {
      MNC_HIP_TRY(hipMemcpyAsync(keep_out, d_keep, (size_t)nk * 4, hipMemcpyDeviceToHost, w->stream));
      MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    }
    *num_out = nk;
  } else {
    // n > 32768: the reference's own arrangement -- bitmask to the host, scan there (nms_kernel.cu:118-140)
    std::vector<u64> hm((size_t)n * cb), remv(cb, 0);
    MNC_HIP_TRY(hipMemcpyAsync(hm.data(), d_mask, (size_t)n * cb * 8, hipMemcpyDeviceToHost, w->stream));
    MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    int nk = 0;
    for (int i = 0; i < n && nk < max_keep; ++i) {
      const int nb = i / 64, ib = i % 64;
      if (!(remv[nb] & (1ULL << ib))) {
        keep_out[nk++] = i;
        const u64* p = hm.data() + (size_t)i * cb;
        for (int j = nb; j < cb; ++j) remv[j] |= p[j];
      }
    }
    *num_out = nk;
  }
  clear_error();
  return MNC_OK;
}

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
            int device_id) {
  MNC_REQUIRE(keep_out && num_out, "mnc_nms: null output pointer");
  return nms_host_impl(keep_out, num_out, nullptr, boxes_host, boxes_num, boxes_dim, thresh, -1, device_id);
}

int mnc_nms_topk(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
                 int max_keep, int device_id) {
  MNC_REQUIRE(keep_out && num_out, "mnc_nms_topk: null output pointer");
  return nms_host_impl(keep_out, num_out, nullptr, boxes_host, boxes_num, boxes_dim, thresh, max_keep, device_id);
}

int mnc_nms_mask(unsigned long long* mask_host, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
                 int device_id) {
  MNC_REQUIRE(mask_host, "mnc_nms_mask: null output pointer");
  return nms_host_impl(nullptr, nullptr, mask_host, boxes_host, boxes_num, boxes_dim, thresh, -1, device_id);
}

void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
          int device_id) {
  if (mnc_nms(keep_out, num_out, boxes_host, boxes_num, boxes_dim, thresh, device_id) != MNC_OK) {
    fprintf(stderr, "mnc_hip: _nms failed: %s\n", mnc_last_error());
    if (num_out) *num_out = 0;
  }
}

}  // extern "C"
