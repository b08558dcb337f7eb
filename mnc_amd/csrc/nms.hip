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
//   * the greedy scan runs on the device in ONE wave per problem (see nms_scan_kernel): the serial dependency is one load
//     latency per block of 64 boxes instead of one per kept box, the intra-block chain runs on the scalar unit, and the
//     4.5 MB mask never crosses PCIe.
//   * batched form (mnc_nms_batched): gpu_mask_voting's 20 per-class NMS problems over one box set run as ONE mask launch
//     (grid.z = class) + ONE scan launch (one wave per class) + one copy each way, instead of 20 synchronous round trips.
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

// Device bitmask layout: COLUMN-BLOCK major, mask[c * n + r] = word of (row box r, column tile c).  Both the mask
// kernel's stores (lane = row within a tile) and the scan's loads (all rows of one column tile) are then contiguous.
//
// grid: (ceil(cb / 4), cb, batch); block: 256.  Item z of the batch uses boxes[order[z*n + i]] as its i-th (sorted)
// box, or boxes[i] when order == nullptr.  mask + z * cb * n is its bitmask.
// n = *n_ptr when n_ptr != nullptr (device-side count, <= n_stride), else n_arg.  n_stride / cb_cap are the strides the
// buffers were sized with (== n / ceil(n/64) in the direct case).
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ order,
                                                       int n_arg, const int* __restrict__ n_ptr, int n_stride, int dim,
                                                       float thr, u64* __restrict__ mask, int cb_cap) {
  const int n = n_ptr ? *n_ptr : n_arg;
  const int cb = (n + 63) >> 6;
  const int rt = blockIdx.y;
  if (rt >= cb) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ct = blockIdx.x * kWavesPerBlock + wave;
  __shared__ float rowbox[64][4];
  if (blockIdx.x * kWavesPerBlock + kWavesPerBlock - 1 < rt) return;  // whole block below the diagonal
  const int* ord = order ? order + (long)blockIdx.z * n_stride : nullptr;
  u64* m = mask + (long)blockIdx.z * cb_cap * n_stride;
  if (threadIdx.x < 64) {
    const int r = rt * 64 + threadIdx.x;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) {
      const float* b = boxes + (long)(ord ? ord[r] : r) * dim;
      v = make_float4(b[0], b[1], b[2], b[3]);
    }
    rowbox[threadIdx.x][0] = v.x; rowbox[threadIdx.x][1] = v.y; rowbox[threadIdx.x][2] = v.z; rowbox[threadIdx.x][3] = v.w;
  }
  __syncthreads();
  if (ct >= cb || ct < rt) return;

  const int c = ct * 64 + lane;
  const bool cvalid = c < n;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  if (cvalid) {
    const float* b = boxes + (long)(ord ? ord[c] : c) * dim;
    b0 = b[0]; b1 = b[1]; b2 = b[2]; b3 = b[3];
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
  if (lane < rows) m[(long)ct * n_stride + rt * 64 + lane] = mine;
}

constexpr int kMaxScanBlocks = 512;  // device scan handles cb <= 512, i.e. n <= 32768
constexpr int kSurvCap = 4096;        // survivor list of the scan (LDS, 8 KB): used when min(max_keep, n) fits

__device__ __forceinline__ u64 uniform64(u64 v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}
// The greedy chain inside one 64-box block (nms_kernel.cu:128-139) on the scalar unit, visiting only the SURVIVORS: a row
// survives iff no earlier survivor suppressed it, so the next survivor is always the lowest row not yet suppressed
// (s_ff1 on the complement of the removal word), and keeping it ORs its word in.  A block of RPN boxes keeps 3-10 of its 64
// rows; stepping through all 64 rows one readlane pair at a time was ~0.8 us of the ~1.6 us a block took.
// diag: lane i holds the word of row i in the block's own column tile; cur: removal word so far (uniform); rows: valid rows.
__device__ __forceinline__ void resolve_block(u64 diag, int rows, int max_keep, u64& cur, u64& kept, int& nk) {
  const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
  u64 live = uniform64(~cur & valid);
  const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
  while (live != 0ull && nk < max_keep) {
    const int i = __builtin_ctzll(live);
    const u64 di = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
    kept |= 1ull << i;
    cur |= di;
    live &= ~di;
    live &= ~(1ull << i);
    ++nk;
  }
}

// grid: batch; block: ONE wave.  Greedy scan of nms_kernel.cu:124-140 on the column-block-major bitmask, stopping after
// max_keep survivors.  Per 64-box block:
//   1. removal word of the block = OR over ALL earlier survivors of their word in this column tile: lane l covers row
//      64*bb + l of every earlier block bb (contiguous, independent loads; survivors' bits come from LDS), then one
//      wave-wide OR reduction -- one load latency per block, not one per survivor;
//   2. the intra-block chain runs on the scalar unit: v_readlane of the 64 diagonal words, 64 unrolled steps;
//   3. survivors write their indices with one compacting vector store (prefix popcount).
// Round 6: the ProposalLayer's tail in the same launch -- g_rois[r] = (0, g_boxes[g_order[keep[r]]]) for r < the survivor count, zero
// rows up to g_cap (what proposal_gather_kernel did as a launch of its own: one of the 57 launches of an image).
__global__ __launch_bounds__(64) void nms_scan_kernel(const u64* __restrict__ mask, int n_arg, const int* __restrict__ n_ptr,
                                                      int n_stride, int cb_cap, int max_keep, int* __restrict__ keep,
                                                      int* __restrict__ num_out, const float* __restrict__ g_boxes = nullptr,
                                                      const int* __restrict__ g_order = nullptr, float* __restrict__ g_rois = nullptr,
                                                      int g_cap = 0) {
  __shared__ u64 s_kept[kMaxScanBlocks];
  __shared__ unsigned short s_surv[kSurvCap];       // rows of the survivors so far (n <= 32768: a row fits in 16 bits)
  const int lane = threadIdx.x;
  const int n = n_ptr ? *n_ptr : n_arg;
  const int cb = (n + 63) >> 6;
  const u64* m = mask + (long)blockIdx.x * cb_cap * n_stride;
  int* kp = keep + (long)blockIdx.x * n_stride;
  int nk = 0;
  // With a bounded survivor count (ProposalLayer: 300 / 1000 of 6000; voting: 100 per class) the removal word of a block is the OR
  // over the LIST of survivors -- ceil(nk / 64) loads per block -- instead of a pass over every earlier block (blk loads per
  // block, cb^2 / 2 = 4400 wave loads at n = 6000).  Same set of words, same result.
  if (min(max_keep, n) <= kSurvCap) {
    // The words of column tile c are requested TWO blocks ahead (while block c - 2 is resolved), so that the load latency
    // (~1 us: the mask was written by other XCDs and comes from Infinity Cache) is not paid once per block: at that time the
    // survivors of blocks < c - 2 are known (list loads), blocks c - 2 and c - 1 are not -- so ALL their rows' words for
    // tile c are fetched (one coalesced 512-byte load each) and a row's word is folded in once its block is resolved and
    // the row turned out to survive.  part* are lane-local partial ORs; the wave-wide OR happens when the tile is consumed.
    auto word = [&](int c, int row) -> u64 { return (c < cb && row < n) ? m[(long)c * n_stride + row] : 0ull; };
    // tile 0: nothing precedes it; tile 1: only block 0 precedes it
    u64 part0 = 0, diag0 = word(0, lane);
    u64 part1 = 0, prev1 = word(1, lane), diag1 = word(1, 64 + lane);        // prev1: rows of block 0 in tile 1
    for (int blk = 0; blk < cb && nk < max_keep; ++blk) {
      // request tile blk + 2: known survivors (blocks < blk), every row of blocks blk and blk + 1, its own diagonal rows
      const int c2 = blk + 2;
      u64 part2 = 0;
      if (c2 < cb)
        for (int s = lane; s < nk; s += 64) part2 |= m[(long)c2 * n_stride + s_surv[s]];
      const u64 rows2a = word(c2, blk * 64 + lane), rows2b = word(c2, (blk + 1) * 64 + lane), diag2 = word(c2, c2 * 64 + lane);
      // consume tile blk
      u64 acc = part0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
      u64 cur = uniform64(acc);
      const int rows = min(n - blk * 64, 64);
      u64 kept = 0;
      const int nk0 = nk;
      resolve_block(diag0, rows, max_keep, cur, kept, nk);
      const bool mine = (kept >> lane) & 1ull;
      if (mine) {
        const int pos = nk0 + __popcll(kept & ((1ull << lane) - 1ull));
        kp[pos] = blk * 64 + lane;
        s_surv[pos] = (unsigned short)(blk * 64 + lane);
      }
      // fold this block's survivors into the two tiles already in flight, then advance the window
      part0 = part1 | (mine ? prev1 : 0ull);
      diag0 = diag1;
      part1 = part2 | (mine ? rows2a : 0ull);
      prev1 = rows2b;
      diag1 = diag2;
      __syncthreads();                               // s_surv of this block is read by the next iteration's list loads
    }
    if (lane == 0) num_out[blockIdx.x] = nk;
    if (g_rois) {                                    // (single-problem launches only; the survivors' rows are in LDS)
      for (int r = lane; r < g_cap; r += 64) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nk) b = reinterpret_cast<const float4*>(g_boxes)[g_order[s_surv[r]]];
        g_rois[r * 5 + 0] = 0.f;
        g_rois[r * 5 + 1] = b.x; g_rois[r * 5 + 2] = b.y; g_rois[r * 5 + 3] = b.z; g_rois[r * 5 + 4] = b.w;
      }
    }
    return;
  }
  // unbounded survivor count (mnc_nms with max_keep < 0 on a large n): one pass over the earlier blocks per block
  for (int blk = 0; blk < cb && nk < max_keep; ++blk) {
    const u64* col = m + (long)blk * n_stride;
    u64 acc = 0;
    for (int bb = 0; bb < blk; ++bb)
      if ((s_kept[bb] >> lane) & 1ull) acc |= col[bb * 64 + lane];
    const int row = blk * 64 + lane;
    const u64 diag = row < n ? col[row] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
    u64 cur = uniform64(acc);
    const int rows = min(n - blk * 64, 64);
    u64 kept = 0;
    const int nk0 = nk;
    resolve_block(diag, rows, max_keep, cur, kept, nk);
    if ((kept >> lane) & 1ull) kp[nk0 + __popcll(kept & ((1ull << lane) - 1ull))] = row;
    if (lane == 0) s_kept[blk] = kept;
    __syncthreads();
  }
  if (lane == 0) num_out[blockIdx.x] = nk;
  if (g_rois) {                                      // (the survivors' rows were written to `keep` by this wave: visible behind the barrier)
    __threadfence_block();
    __syncthreads();
    for (int r = lane; r < g_cap; r += 64) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nk) b = reinterpret_cast<const float4*>(g_boxes)[g_order[kp[r]]];
      g_rois[r * 5 + 0] = 0.f;
      g_rois[r * 5 + 1] = b.x; g_rois[r * 5 + 2] = b.y; g_rois[r * 5 + 3] = b.z; g_rois[r * 5 + 4] = b.w;
    }
  }
}

// ---- launchers ------------------------------------------------------------------------------------------------
int nms_mask_launch(hipStream_t stream, const float* d_boxes, const int* d_order, int n, int dim, float thr, u64* d_mask,
                    int batch) {
  const int cb = cdiv(n, 64);
  dim3 grid(cdiv(cb, kWavesPerBlock), cb, batch);
  hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, stream, d_boxes, d_order, n, (const int*)nullptr, n, dim, thr,
                     d_mask, cb);
  return MNC_OK;
}

int nms_mask_launch_indirect(hipStream_t stream, const float* d_boxes, const int* d_order, const int* d_n, int n_cap,
                             int dim, float thr, u64* d_mask) {
  const int cb = cdiv(n_cap, 64);
  dim3 grid(cdiv(cb, kWavesPerBlock), cb, 1);
  hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, stream, d_boxes, d_order, 0, d_n, n_cap, dim, thr, d_mask, cb);
  return MNC_OK;
}

int nms_scan_launch(hipStream_t stream, const u64* d_mask, int n, int max_keep, int* d_keep, int* d_num, int batch) {
  const int cb = cdiv(n, 64);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, stream, d_mask, n, (const int*)nullptr, n, cb, max_keep,
                     d_keep, d_num);
  return MNC_OK;
}

int nms_scan_launch_indirect(hipStream_t stream, const u64* d_mask, const int* d_n, int n_cap, int max_keep, int* d_keep,
                             int* d_num, const float* d_gather_boxes, const int* d_gather_order, float* d_rois, int rois_cap) {
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, stream, d_mask, 0, d_n, n_cap, cdiv(n_cap, 64), max_keep, d_keep,
                     d_num, d_gather_boxes, d_gather_order, d_rois, rois_cap);
  return MNC_OK;
}

// ---- per-device workspace for the host-pointer entry points (b1/b2 allocate-per-call in the reference) ----------
static LegacyWs g_ws[16];

int legacy_ws(int device_id, size_t bytes, LegacyWs** out, std::unique_lock<std::mutex>* lock) {
  int ndev = 0;
  MNC_HIP_TRY(hipGetDeviceCount(&ndev));
  MNC_REQUIRE(device_id >= 0 && device_id < ndev && device_id < 16, "device %d out of range (have %d)", device_id, ndev);
  MNC_HIP_TRY(hipSetDevice(device_id));
  LegacyWs* w = &g_ws[device_id];
  *lock = std::unique_lock<std::mutex>(w->mu);        // before anything is created, grown or freed
  if (!w->stream) MNC_HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
  if (bytes > w->cap) {
    if (w->buf) {
      MNC_HIP_TRY(hipStreamSynchronize(w->stream));
      MNC_HIP_TRY(hipFree(w->buf));
    }
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

// boxes_host [n][dim]; order_host: nullptr (boxes already sorted, batch == 1) or [batch][n] indices into boxes.
// keep_out [batch][n] positions in each item's order; num_out [batch].  mask_out (batch == 1 only): row-major [n][cb].
static int nms_host_impl(int* keep_out, int* num_out, u64* mask_out, const float* boxes_host, int n, int dim,
                         const int* order_host, int batch, float thr, int max_keep, int device_id) {
  MNC_REQUIRE(n >= 0 && dim >= 4 && batch >= 1, "mnc_nms: boxes_num=%d boxes_dim=%d batch=%d", n, dim, batch);
  if (num_out) for (int b = 0; b < batch; ++b) num_out[b] = 0;
  if (n == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(boxes_host, "mnc_nms: null boxes");
  if (order_host)
    for (long i = 0; i < (long)batch * n; ++i)
      MNC_REQUIRE(order_host[i] >= 0 && order_host[i] < n, "mnc_nms_batched: order[%ld]=%d out of range", i, order_host[i]);
  const int cb = cdiv(n, 64);
  if (max_keep < 0 || max_keep > n) max_keep = n;
  const size_t box_b = align256((size_t)n * dim * 4), ord_b = order_host ? align256((size_t)batch * n * 4) : 0;
  const size_t mask_b = align256((size_t)batch * n * cb * 8), keep_b = align256((size_t)batch * n * 4);
  LegacyWs* w = nullptr;
  std::unique_lock<std::mutex> lock;
  int rc = legacy_ws(device_id, box_b + ord_b + mask_b + keep_b + align256((size_t)batch * 4), &w, &lock);
  if (rc) return rc;
  char* base = (char*)w->buf;
  float* d_boxes = (float*)base;
  int* d_order = order_host ? (int*)(base + box_b) : nullptr;
  u64* d_mask = (u64*)(base + box_b + ord_b);
  int* d_keep = (int*)(base + box_b + ord_b + mask_b);
  int* d_num = (int*)(base + box_b + ord_b + mask_b + keep_b);
  MNC_HIP_TRY(hipMemcpyAsync(d_boxes, boxes_host, (size_t)n * dim * 4, hipMemcpyHostToDevice, w->stream));
  if (order_host) MNC_HIP_TRY(hipMemcpyAsync(d_order, order_host, (size_t)batch * n * 4, hipMemcpyHostToDevice, w->stream));
  const bool device_scan = cb <= kMaxScanBlocks && !mask_out;
  if (!device_scan) MNC_HIP_TRY(hipMemsetAsync(d_mask, 0, (size_t)batch * n * cb * 8, w->stream));
  nms_mask_launch(w->stream, d_boxes, d_order, n, dim, thr, d_mask, batch);
  MNC_HIP_TRY(hipGetLastError());
  if (device_scan) {
    nms_scan_launch(w->stream, d_mask, n, max_keep, d_keep, d_num, batch);
    MNC_HIP_TRY(hipGetLastError());
    MNC_HIP_TRY(hipMemcpyAsync(num_out, d_num, (size_t)batch * 4, hipMemcpyDeviceToHost, w->stream));
    if (batch > 1) {  // one copy of the whole (small) keep table instead of a second dependent round trip
      MNC_HIP_TRY(hipMemcpyAsync(keep_out, d_keep, (size_t)batch * n * 4, hipMemcpyDeviceToHost, w->stream));
      MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    } else {
      MNC_HIP_TRY(hipStreamSynchronize(w->stream));
      if (num_out[0] > 0) {
        MNC_HIP_TRY(hipMemcpyAsync(keep_out, d_keep, (size_t)num_out[0] * 4, hipMemcpyDeviceToHost, w->stream));
        MNC_HIP_TRY(hipStreamSynchronize(w->stream));
      }
    }
    clear_error();
    return MNC_OK;
  }
  // mask read-back (parity API) or n > 32768: the reference's own arrangement -- bitmask to the host, scan there
  // (nms_kernel.cu:118-140).  The device layout is column-block major; rows are re-assembled here.
  std::vector<u64> hm((size_t)n * cb);
  for (int b = 0; b < batch; ++b) {
    MNC_HIP_TRY(hipMemcpyAsync(hm.data(), d_mask + (size_t)b * n * cb, (size_t)n * cb * 8, hipMemcpyDeviceToHost, w->stream));
    MNC_HIP_TRY(hipStreamSynchronize(w->stream));
    if (mask_out) {
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < cb; ++c) mask_out[(size_t)r * cb + c] = hm[(size_t)c * n + r];
      continue;
    }
    std::vector<u64> remv(cb, 0);
    int nk = 0;
    int* kp = keep_out + (size_t)b * n;
    for (int i = 0; i < n && nk < max_keep; ++i) {
      const int nb = i / 64, ib = i % 64;
      if (!(remv[nb] & (1ULL << ib))) {
        kp[nk++] = i;
        for (int j = nb; j < cb; ++j) remv[j] |= hm[(size_t)j * n + i];
      }
    }
    num_out[b] = nk;
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
  return nms_host_impl(keep_out, num_out, nullptr, boxes_host, boxes_num, boxes_dim, nullptr, 1, thresh, -1, device_id);
}

int mnc_nms_batched(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                    const int* order_host, int batch, float thresh, int max_keep, int device_id) {
  MNC_REQUIRE(keep_out && num_out && (boxes_num == 0 || order_host), "mnc_nms_batched: null pointer");
  return nms_host_impl(keep_out, num_out, nullptr, boxes_host, boxes_num, boxes_dim, order_host, batch, thresh, max_keep,
                       device_id);
}

int mnc_nms_topk(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
                 int max_keep, int device_id) {
  MNC_REQUIRE(keep_out && num_out, "mnc_nms_topk: null output pointer");
  return nms_host_impl(keep_out, num_out, nullptr, boxes_host, boxes_num, boxes_dim, nullptr, 1, thresh, max_keep, device_id);
}

int mnc_nms_mask(unsigned long long* mask_host, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
                 int device_id) {
  MNC_REQUIRE(mask_host, "mnc_nms_mask: null output pointer");
  return nms_host_impl(nullptr, nullptr, mask_host, boxes_host, boxes_num, boxes_dim, nullptr, 1, thresh, -1, device_id);
}

void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float thresh,
          int device_id) {
  if (mnc_nms(keep_out, num_out, boxes_host, boxes_num, boxes_dim, thresh, device_id) != MNC_OK) {
    fprintf(stderr, "mnc_hip: _nms failed: %s\n", mnc_last_error());
    if (num_out) *num_out = 0;
  }
}

}  // extern "C"
