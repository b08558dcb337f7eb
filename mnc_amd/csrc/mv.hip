// Mask voting for gfx950 -- replaces lib/nms/mv_kernel.cu (reference: 7 kernels, an N*H*W float render buffer
// -- 1.44 GB at 600 masks on a 600x1000 canvas -- an R*H*W aggregate buffer and 12 cudaMalloc/cudaFree per call).
//
// Fused design: the aggregate  A_r(h,w) = sum_{i in cand(r)} render_i(h,w) * weight_i  (mask_render :36-91 +
// mask_aggregate :93-112) is a pure function of (r,h,w), so it is never stored:
//   kernel 1  mv_bounds   : grid (R, splits).  Each block walks a slab of the union box of result r's candidate boxes
//                           (outside it every render is 0, and 0 > 0.4 is false), evaluates A_r per pixel with the
//                           candidate boxes/weights staged in LDS, and reduces "A_r > 0.4" to min/max x,y with
//                           wave shuffles + one atomicMin/Max per block  (reduce_mask_col/row + reduce_bounding_x/y,
//                           :114-190).
//   kernel 2  mv_resample : grid R, 448 threads.  Re-evaluates A_r at the <= 4 integer pixels each of the 21x21
//                           output samples needs (mask_resize :193-240).
// HBM traffic is the algorithmic minimum: masks + boxes in (1.06 MB at N=600), R*441 floats + R*4 ints out.
//
// Compiled with -ffp-contract=off; every float expression is written in the reference's operation order, so the
// outputs are bit-exact with mv_kernel.cu evaluated without FMA contraction (== oracle/mnc_oracle.c == oracle/_ref).
#include <algorithm>
#include <climits>
#include <functional>
#include <mutex>

#include <chrono>
#include <cstdlib>

#include "mnc_internal.h"

namespace mnc {

constexpr int kMaxOrderDevice = 4096;   // boxes per class the LDS bitonic sort orders on the device (32 KB of keys)
constexpr int kMaxCandLds = 1024;  // candidate descriptors staged in LDS per result (6 KB x 4); more -> chunked

struct CandLds {
  float x1[kMaxCandLds], y1[kMaxCandLds], x2[kMaxCandLds], y2[kMaxCandLds], w[kMaxCandLds];
  int m[kMaxCandLds];
};

// mask_render for one pixel of one mask, mv_kernel.cu:36-91 (operation order preserved).
__device__ __forceinline__ float render_px(float x1, float y1, float x2, float y2, const float* __restrict__ mask, int S,
                                           int h, int w) {
  if (w < x1 || w > x2 || h < y1 || h > y2) return 0.0f;
  const float bw = (float)((double)(x2 - x1) + 1.0);
  const float bh = (float)((double)(y2 - y1) + 1.0);
  const float rw = (float)S / bw, rh = (float)S / bh;
  const float ix = ((float)w - x1) * rw, iy = ((float)h - y1) * rh;
  const int sx = (int)floorf(ix), sy = (int)floorf(iy);
  if (sx == S - 1 || sy == S - 1) return mask[sy * S + sx];
  const int tl = sy * S + sx, tr = tl + 1, bl = tl + S, br = bl + 1;
  const float fx = ix - sx, fy = iy - sy;
  const float wtl = (1 - fx) * (1 - fy), wtr = fx * (1 - fy), wbl = (1 - fx) * fy, wbr = fx * fy;
  return wtl * mask[tl] + wtr * mask[tr] + wbl * mask[bl] + wbr * mask[br];
}

// A_r(h, w): candidates in list order, val += render * weight (mask_aggregate, mv_kernel.cu:104-110).
__device__ __forceinline__ float aggregate_px(const CandLds& cl, int nc, const float* __restrict__ masks, int S, int h,
                                              int w) {
  float val = 0.0f;
  for (int i = 0; i < nc; ++i) {
    const float r = render_px(cl.x1[i], cl.y1[i], cl.x2[i], cl.y2[i], masks + (long)cl.m[i] * S * S, S, h, w);
    val += r * cl.w[i];
  }
  return val;
}

// Slow path for results with more than kMaxCandLds candidates: descriptors straight from global memory.
__device__ float aggregate_px_global(const float* __restrict__ boxes, int box_dim, const float* __restrict__ masks,
                                     const int* __restrict__ inds, const float* __restrict__ wts, int c0, int c1, int S,
                                     int h, int w) {
  float val = 0.0f;
  for (int i = c0; i < c1; ++i) {
    const int m = inds[i];
    const float* b = boxes + (long)m * box_dim;
    val += render_px(b[0], b[1], b[2], b[3], masks + (long)m * S * S, S, h, w) * wts[i];
  }
  return val;
}

__device__ __forceinline__ void stage_cands(CandLds& cl, const float* __restrict__ boxes, int box_dim,
                                            const int* __restrict__ inds, const float* __restrict__ wts, int c0, int nc) {
  for (int i = threadIdx.x; i < nc; i += blockDim.x) {
    const int m = inds[c0 + i];
    const float* b = boxes + (long)m * box_dim;
    cl.x1[i] = b[0]; cl.y1[i] = b[1]; cl.x2[i] = b[2]; cl.y2[i] = b[3];
    cl.w[i] = wts[c0 + i];
    cl.m[i] = m;
  }
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// bounds: [R][4] = (min x, min y, max x, max y), pre-set to (INT_MAX, INT_MAX, -1, -1).
__global__ __launch_bounds__(256) void mv_bounds_kernel(const float* __restrict__ boxes, int box_dim,
                                                        const float* __restrict__ masks, int S,
                                                        const int* __restrict__ inds, const int* __restrict__ begins,
                                                        const int* __restrict__ ends, const float* __restrict__ wts,
                                                        int H, int W, int* __restrict__ bounds) {
  __shared__ CandLds cl;
  __shared__ int red[4][4];
  __shared__ int ubox[4];
  const int r = blockIdx.x;
  const int c0 = begins[r], c1 = ends[r];
  const int nc = c1 - c0;
  if (nc <= 0) return;
  const bool in_lds = nc <= kMaxCandLds;
  if (in_lds) stage_cands(cl, boxes, box_dim, inds, wts, c0, nc);
  if (threadIdx.x == 0) { ubox[0] = INT_MAX; ubox[1] = INT_MAX; ubox[2] = -1; ubox[3] = -1; }
  __syncthreads();
  // union of the candidates' pixel extents: pixel w is inside box iff x1 <= w <= x2 (float compare, :52)
  {
    int lx = INT_MAX, ly = INT_MAX, hx = -1, hy = -1;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
      float x1, y1, x2, y2;
      if (in_lds) { x1 = cl.x1[i]; y1 = cl.y1[i]; x2 = cl.x2[i]; y2 = cl.y2[i]; }
      else { const float* b = boxes + (long)inds[c0 + i] * box_dim; x1 = b[0]; y1 = b[1]; x2 = b[2]; y2 = b[3]; }
      // conservative integer hull, clipped to the canvas
      const int ax = max(0, (int)floorf(x1)), ay = max(0, (int)floorf(y1));
      const int bx = min(W - 1, (int)ceilf(x2)), by = min(H - 1, (int)ceilf(y2));
      if (ax <= bx && ay <= by) { lx = min(lx, ax); ly = min(ly, ay); hx = max(hx, bx); hy = max(hy, by); }
    }
    lx = wave_min(lx); ly = wave_min(ly); hx = wave_max(hx); hy = wave_max(hy);
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&ubox[0], lx); atomicMin(&ubox[1], ly); atomicMax(&ubox[2], hx); atomicMax(&ubox[3], hy);
    }
  }
  __syncthreads();
  const int ux1 = ubox[0], uy1 = ubox[1], ux2 = ubox[2], uy2 = ubox[3];
  if (ux2 < ux1 || uy2 < uy1) return;
  const int uw = ux2 - ux1 + 1, uh = uy2 - uy1 + 1;
  // this block's slab of rows
  const int rows_per = (uh + gridDim.y - 1) / gridDim.y;
  const int ya = uy1 + blockIdx.y * rows_per, yb = min(uy2 + 1, ya + rows_per);
  int lx = INT_MAX, ly = INT_MAX, hx = -1, hy = -1;
  const long npx = (long)uw * max(0, yb - ya);
  for (long p = threadIdx.x; p < npx; p += blockDim.x) {
    const int h = ya + (int)(p / uw), w = ux1 + (int)(p % uw);
    const float v = in_lds ? aggregate_px(cl, nc, masks, S, h, w)
                           : aggregate_px_global(boxes, box_dim, masks, inds, wts, c0, c1, S, h, w);
    if (v > 0.4f) {  // BINARIZE_THRESH, strict (mv_kernel.cu:13, :121, :136)
      lx = min(lx, w); hx = max(hx, w); ly = min(ly, h); hy = max(hy, h);
    }
  }
  lx = wave_min(lx); ly = wave_min(ly); hx = wave_max(hx); hy = wave_max(hy);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = lx; red[wave][1] = ly; red[wave][2] = hx; red[wave][3] = hy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      lx = min(lx, red[k][0]); ly = min(ly, red[k][1]); hx = max(hx, red[k][2]); hy = max(hy, red[k][3]);
    }
    if (hx >= 0) {
      atomicMin(&bounds[r * 4 + 0], lx); atomicMin(&bounds[r * 4 + 1], ly);
      atomicMax(&bounds[r * 4 + 2], hx); atomicMax(&bounds[r * 4 + 3], hy);
    }
  }
}

// grid R, block 448 (7 waves; 441 active).  Finalises the box (defaults W/2, H/2, :149,:173) and resamples (:193-240).
__global__ __launch_bounds__(448) void mv_resample_kernel(const float* __restrict__ boxes, int box_dim,
                                                          const float* __restrict__ masks, int S,
                                                          const int* __restrict__ inds, const int* __restrict__ begins,
                                                          const int* __restrict__ ends, const float* __restrict__ wts,
                                                          int H, int W, const int* __restrict__ bounds,
                                                          float* __restrict__ out_mask, int* __restrict__ out_box) {
  __shared__ CandLds cl;
  const int r = blockIdx.x;
  const int c0 = begins[r], c1 = ends[r];
  const int nc = max(c1 - c0, 0);
  const bool in_lds = nc <= kMaxCandLds;
  if (in_lds) stage_cands(cl, boxes, box_dim, inds, wts, c0, nc);
  __syncthreads();
  int bx1 = bounds[r * 4 + 0], by1 = bounds[r * 4 + 1], bx2 = bounds[r * 4 + 2], by2 = bounds[r * 4 + 3];
  if (bx2 < 0) { bx1 = W / 2; bx2 = W / 2; }   // no column reached 0.4
  if (by2 < 0) { by1 = H / 2; by2 = H / 2; }
  if (threadIdx.x == 0) {
    out_box[r * 4 + 0] = bx1; out_box[r * 4 + 1] = by1; out_box[r * 4 + 2] = bx2; out_box[r * 4 + 3] = by2;
  }
  for (int idx = threadIdx.x; idx < S * S; idx += blockDim.x) {
    const int w = idx % S, h = idx / S;
    const float bw = (float)((double)(bx2 - bx1) + 1.0), bh = (float)((double)(by2 - by1) + 1.0);
    const float rw = bw / (float)S, rh = bh / (float)S;
    const float ix = bx1 + (float)w * rw, iy = by1 + (float)h * rh;
    const int sx = (int)floorf(ix), sy = (int)floorf(iy);
#define MNC_AGG(hh, ww) (in_lds ? aggregate_px(cl, nc, masks, S, (hh), (ww)) \
                                : aggregate_px_global(boxes, box_dim, masks, inds, wts, c0, c1, S, (hh), (ww)))
    float v;
    if (sx == W - 1 || sy == H - 1) {
      v = MNC_AGG(sy, sx);
    } else {
      const float fx = ix - sx, fy = iy - sy;
      const float wtl = (1 - fx) * (1 - fy), wtr = fx * (1 - fy), wbl = (1 - fx) * fy, wbr = fx * fy;
      const float atl = MNC_AGG(sy, sx), atr = MNC_AGG(sy, sx + 1), abl = MNC_AGG(sy + 1, sx), abr = MNC_AGG(sy + 1, sx + 1);
      v = wtl * atl + wtr * atr + wbl * abl + wbr * abr;
    }
#undef MNC_AGG
    out_mask[((long)r * S + h) * S + w] = v;
  }
}

__global__ void mv_init_bounds_kernel(int* bounds, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * 4) bounds[i] = (i & 3) < 2 ? INT_MAX : -1;
}

// ---- device-side preparation of the voting problem (mask_transform.py:228-270) ------------------------------------
// Per-class descending score order == np.argsort(-scores[:, c], kind="stable"): one workgroup per class sorts 64-bit keys
// (orderable(-score) << 32 | index) with a bitonic network in LDS.  -0/+0 are one value (ties -> index order) and NaN sorts
// last, as in numpy.
__global__ __launch_bounds__(1024) void mv_order_kernel(const float* __restrict__ scores, int n, int C, int NP,
                                                        int* __restrict__ order) {
  extern __shared__ unsigned long long s_keys[];
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const float v = -scores[(long)i * C + c + 1] + 0.0f;
      const unsigned u = __float_as_uint(v);
      const unsigned o = (v != v) ? 0xFFFFFFFFu : (u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u));
      k = ((unsigned long long)o << 32) | (unsigned)i;
    }
    s_keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        const int x = i ^ j;
        if (x > i) {
          const unsigned long long a = s_keys[i], b = s_keys[x];
          if ((a > b) == ((i & k) == 0)) { s_keys[i] = b; s_keys[x] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[(long)c * n + i] = (int)(s_keys[i] & 0xFFFFFFFFu);
}

// keep lists index the sorted order; the host wants box indices: keepbox[c][k] = order[c][keep[c][k]]
__global__ void mv_keepbox_kernel(const int* __restrict__ order, const int* __restrict__ keep, const int* __restrict__ num,
                                  int n, int* __restrict__ keepbox) {
  const int c = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < num[c]) keepbox[(long)c * n + k] = order[(long)c * n + keep[(long)c * n + k]];
}

// Candidate set of result row r = (kept box bi, class c): members {i : IoU_f64(box_i, box_bi) >= iou_thresh} in index order,
// weights = class scores normalised by python's sequential float32 sum (mask_transform.py:253-270).  One wave per row;
// row r owns cinds/cw[r*n .. r*n + n).
__global__ __launch_bounds__(64) void mv_candidates_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           int n, int C, const int* __restrict__ rows, float iou_thresh,
                                                           int* __restrict__ cinds, float* __restrict__ cw,
                                                           int* __restrict__ cbegin, int* __restrict__ cend) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const int bi = rows[2 * r], c = rows[2 * r + 1];
  const double q0 = boxes[bi * 4 + 0], q1 = boxes[bi * 4 + 1], q2 = boxes[bi * 4 + 2], q3 = boxes[bi * 4 + 3];
  const double qarea = (q2 - q0 + 1) * (q3 - q1 + 1);
  int* my_inds = cinds + (long)r * n;
  float* my_w = cw + (long)r * n;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    bool member = false;
    if (i < n) {
      const double b0 = boxes[i * 4 + 0], b1 = boxes[i * 4 + 1], b2 = boxes[i * 4 + 2], b3 = boxes[i * 4 + 3];
      double ov = 0.0;
      const double iw = (b2 < q2 ? b2 : q2) - (b0 > q0 ? b0 : q0) + 1;
      if (iw > 0) {
        const double ih = (b3 < q3 ? b3 : q3) - (b1 > q1 ? b1 : q1) + 1;
        if (ih > 0) ov = iw * ih / ((b2 - b0 + 1) * (b3 - b1 + 1) + qarea - iw * ih);
      }
      member = ov >= (double)iou_thresh;
    }
    const unsigned long long bal = __ballot(member);
    if (member) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      my_inds[pos] = i;
      my_w[pos] = scores[(long)i * C + c + 1];
    }
    base += __popcll(bal);
  }
  __syncthreads();
  float sum = 0.0f;
  if (lane == 0)
    for (int t = 0; t < base; ++t) sum = t ? sum + my_w[t] : my_w[t];      // ((0 + w0) + w1) + ...  (0 + w0 is exact)
  sum = __shfl(sum, 0);
  __syncthreads();
  for (int t = lane; t < base; t += 64) my_w[t] = my_w[t] / sum;
  if (lane == 0) {
    cbegin[r] = r * n;
    cend[r] = r * n + base;
  }
}

// All pointers device.  d_bounds: R*4 ints scratch.  Result r's candidates are d_inds/d_wts[d_begins[r] .. d_ends[r]).
int mv_launch(hipStream_t stream, const float* d_boxes, int box_dim, const float* d_masks, int S, const int* d_inds,
              const int* d_begins, const int* d_ends, const float* d_wts, int H, int W, int R, int* d_bounds,
              float* d_out_mask, int* d_out_box) {
  if (R <= 0) return MNC_OK;
  hipLaunchKernelGGL(mv_init_bounds_kernel, dim3(cdiv(R * 4, 256)), dim3(256), 0, stream, d_bounds, R);
  // ~2048 blocks in flight: R results x `splits` row slabs each
  int splits = 2048 / R;
  if (splits < 1) splits = 1;
  if (splits > 32) splits = 32;
  hipLaunchKernelGGL(mv_bounds_kernel, dim3(R, splits), dim3(256), 0, stream, d_boxes, box_dim, d_masks, S, d_inds,
                     d_begins, d_ends, d_wts, H, W, d_bounds);
  hipLaunchKernelGGL(mv_resample_kernel, dim3(R), dim3(448), 0, stream, d_boxes, box_dim, d_masks, S, d_inds, d_begins,
                     d_ends, d_wts, H, W, d_bounds, d_out_mask, d_out_box);
  return MNC_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace mnc

using namespace mnc;

extern "C" {

int mnc_mv(const float* all_boxes, const float* all_masks, int all_boxes_num, const int* candidate_inds,
           const int* candidate_start, const float* candidate_weights, int candidate_num, int image_height,
           int image_width, int box_dim, int mask_size, int result_num, float* out_mask, int* out_box, int device_id) {
  MNC_REQUIRE(all_boxes_num >= 0 && candidate_num >= 0 && result_num >= 0, "mnc_mv: negative count");
  MNC_REQUIRE(box_dim >= 4 && mask_size >= 2 && image_height > 0 && image_width > 0,
              "mnc_mv: box_dim=%d mask_size=%d H=%d W=%d", box_dim, mask_size, image_height, image_width);
  if (result_num == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(out_mask && out_box && candidate_start, "mnc_mv: null pointer");
  MNC_REQUIRE(candidate_num == 0 || (all_boxes && all_masks && candidate_inds && candidate_weights), "mnc_mv: null pointer");
  for (int r = 0; r < result_num; ++r) {
    const int a = r ? candidate_start[r - 1] : 0, b = candidate_start[r];
    MNC_REQUIRE(a <= b && b <= candidate_num, "mnc_mv: candidate_start[%d]=%d not a monotone END offset <= %d", r, b,
                candidate_num);
  }
  for (int i = 0; i < candidate_num; ++i)
    MNC_REQUIRE(candidate_inds[i] >= 0 && candidate_inds[i] < all_boxes_num, "mnc_mv: candidate_inds[%d]=%d out of range",
                i, candidate_inds[i]);
  const int S = mask_size, R = result_num;
  const size_t b_boxes = align256((size_t)all_boxes_num * box_dim * 4), b_masks = align256((size_t)all_boxes_num * S * S * 4);
  const size_t b_inds = align256((size_t)candidate_num * 4), b_wts = b_inds, b_starts = align256((size_t)R * 8);
  const size_t b_bounds = align256((size_t)R * 16), b_omask = align256((size_t)R * S * S * 4), b_obox = align256((size_t)R * 16);
  LegacyWs* w = nullptr;
  int rc = legacy_ws(device_id, b_boxes + b_masks + b_inds + b_wts + b_starts + b_bounds + b_omask + b_obox, &w);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(w->mu);
  char* p = (char*)w->buf;
  float* d_boxes = (float*)p; p += b_boxes;
  float* d_masks = (float*)p; p += b_masks;
  int* d_inds = (int*)p; p += b_inds;
  float* d_wts = (float*)p; p += b_wts;
  int* d_starts = (int*)p; p += b_starts;
  int* d_bounds = (int*)p; p += b_bounds;
  float* d_omask = (float*)p; p += b_omask;
  int* d_obox = (int*)p;
  hipStream_t s = w->stream;
  if (all_boxes_num) {
    MNC_HIP_TRY(hipMemcpyAsync(d_boxes, all_boxes, (size_t)all_boxes_num * box_dim * 4, hipMemcpyHostToDevice, s));
    MNC_HIP_TRY(hipMemcpyAsync(d_masks, all_masks, (size_t)all_boxes_num * S * S * 4, hipMemcpyHostToDevice, s));
  }
  if (candidate_num) {
    MNC_HIP_TRY(hipMemcpyAsync(d_inds, candidate_inds, (size_t)candidate_num * 4, hipMemcpyHostToDevice, s));
    MNC_HIP_TRY(hipMemcpyAsync(d_wts, candidate_weights, (size_t)candidate_num * 4, hipMemcpyHostToDevice, s));
  }
  std::vector<int> h_begins(R);              // candidate_start holds END offsets (gpu_mv.pyx / mv_kernel.cu:100)
  for (int r = 0; r < R; ++r) h_begins[r] = r ? candidate_start[r - 1] : 0;
  MNC_HIP_TRY(hipMemcpyAsync(d_starts, h_begins.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
  MNC_HIP_TRY(hipMemcpyAsync(d_starts + R, candidate_start, (size_t)R * 4, hipMemcpyHostToDevice, s));
  mv_launch(s, d_boxes, box_dim, d_masks, S, d_inds, d_starts, d_starts + R, d_wts, image_height, image_width, R, d_bounds,
            d_omask, d_obox);
  MNC_HIP_TRY(hipGetLastError());
  MNC_HIP_TRY(hipMemcpyAsync(out_mask, d_omask, (size_t)R * S * S * 4, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipMemcpyAsync(out_box, d_obox, (size_t)R * 16, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipStreamSynchronize(s));
  clear_error();
  return MNC_OK;
}

// gpu_mask_voting in one call (lib/transform/mask_transform.py:213-286 + lib/nms/mv_kernel.cu).  See include/mnc_hip.h.
// on_device: boxes / masks / scores are device pointers (the engine's own outputs) and `ext_stream` is the stream they were
// produced on; otherwise they are host pointers and the library's per-device stream is used.
static int mask_voting_core(bool on_device, hipStream_t ext_stream, const float* boxes, const float* masks,
                            const float* scores, const int* order, int n, int num_classes, int mask_size, int max_per_image,
                            float nms_thresh, float iou_thresh, int image_height, int image_width, float* out_mask,
                            int* out_box, float* out_score, int* class_count, int* result_num, int device_id) {
  MNC_REQUIRE(result_num && class_count, "mnc_mask_voting: null output pointer");
  MNC_REQUIRE(n >= 0 && num_classes >= 2 && mask_size >= 2 && max_per_image > 0 && image_height > 0 && image_width > 0,
              "mnc_mask_voting: bad argument");
  const int B = num_classes - 1, S = mask_size;
  *result_num = 0;
  for (int c = 0; c < B; ++c) class_count[c] = 0;
  if (n == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(boxes && masks && scores && out_mask && out_box && out_score, "mnc_mask_voting: null pointer");
  if (order)
    for (long i = 0; i < (long)B * n; ++i)
      MNC_REQUIRE(order[i] >= 0 && order[i] < n, "mnc_mask_voting: order[%ld]=%d out of range", i, order[i]);
  const int cb = cdiv(n, 64);
  const int keep_cap = max_per_image < n ? max_per_image : n;
  const int Rmax = B * keep_cap;
  const size_t b_boxes = align256((size_t)n * 16), b_masks = align256((size_t)n * S * S * 4), b_order = align256((size_t)B * n * 4);
  const size_t b_scores = align256((size_t)n * num_classes * 4);
  const size_t b_bits = align256((size_t)B * n * cb * 8), b_keep = align256((size_t)B * n * 4), b_num = align256((size_t)B * 4);
  const size_t b_cinds = align256((size_t)Rmax * n * 4), b_cw = b_cinds, b_rows = align256((size_t)Rmax * 8);
  const size_t b_cse = align256((size_t)Rmax * 4);
  const size_t b_bounds = align256((size_t)Rmax * 16), b_omask = align256((size_t)Rmax * S * S * 4), b_obox = b_bounds;
  LegacyWs* w = nullptr;
  int rc = legacy_ws(device_id, b_boxes + b_masks + b_order + b_scores + b_bits + 2 * b_keep + b_num + b_cinds + b_cw + b_rows +
                                    2 * b_cse + b_bounds + b_omask + b_obox, &w);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(w->mu);
  hipStream_t s = on_device ? ext_stream : w->stream;
  char* p = (char*)w->buf;
  const float* d_boxes = on_device ? boxes : (const float*)p; p += b_boxes;
  const float* d_masks = on_device ? masks : (const float*)p; p += b_masks;
  int* d_order = (int*)p; p += b_order;
  const float* d_scores = on_device ? scores : (const float*)p; p += b_scores;
  unsigned long long* d_bits = (unsigned long long*)p; p += b_bits;
  int* d_keep = (int*)p; p += b_keep;
  int* d_keepbox = (int*)p; p += b_keep;
  int* d_num = (int*)p; p += b_num;
  int* d_cinds = (int*)p; p += b_cinds;
  float* d_cw = (float*)p; p += b_cw;
  int* d_rows = (int*)p; p += b_rows;
  int* d_cbegin = (int*)p; p += b_cse;
  int* d_cend = (int*)p; p += b_cse;
  int* d_bounds = (int*)p; p += b_bounds;
  float* d_omask = (float*)p; p += b_omask;
  int* d_obox = (int*)p;

  const bool timing = getenv("MNC_MV_TIMING") != nullptr;      // diagnostic: host wall-clock of each phase on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3;
  };
  const auto t0 = now();
  // 1. per-class score order + the per-class NMS problems (mask_transform.py:228-240), batched; kept lists come back as
  //    box indices; the masks ride along on the same stream
  std::vector<float> h_scores;                 // the host picks the threshold and the result rows from the class scores
  if (on_device) {
    h_scores.resize((size_t)n * num_classes);
    MNC_HIP_TRY(hipMemcpyAsync(h_scores.data(), d_scores, h_scores.size() * 4, hipMemcpyDeviceToHost, s));
    scores = h_scores.data();                  // valid after the synchronisation below
  } else {
    MNC_HIP_TRY(hipMemcpyAsync((void*)d_boxes, boxes, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MNC_HIP_TRY(hipMemcpyAsync((void*)d_scores, scores, (size_t)n * num_classes * 4, hipMemcpyHostToDevice, s));
  }
  std::vector<int> h_order;
  if (order) {
    MNC_HIP_TRY(hipMemcpyAsync(d_order, order, (size_t)B * n * 4, hipMemcpyHostToDevice, s));
  } else if (n <= kMaxOrderDevice) {
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(mv_order_kernel, dim3(B), dim3(np2 < 1024 ? np2 : 1024), (size_t)np2 * 8, s, d_scores, n, num_classes,
                       np2, d_order);
  } else {                                 // larger than the LDS sort: the same order on the host
    if (on_device) MNC_HIP_TRY(hipStreamSynchronize(s));
    h_order.resize((size_t)B * n);
    for (int c = 0; c < B; ++c) {
      int* o = h_order.data() + (size_t)c * n;
      for (int i = 0; i < n; ++i) o[i] = i;
      const float* sc = scores + c + 1;
      std::stable_sort(o, o + n, [&](int a, int b) {
        const float va = -sc[(size_t)a * num_classes], vb = -sc[(size_t)b * num_classes];
        return va < vb || (vb != vb && va == va);              // NaN last, as numpy
      });
    }
    MNC_HIP_TRY(hipMemcpyAsync(d_order, h_order.data(), (size_t)B * n * 4, hipMemcpyHostToDevice, s));
  }
  nms_mask_launch(s, d_boxes, d_order, n, 4, nms_thresh, d_bits, B);
  nms_scan_launch(s, d_bits, n, keep_cap, d_keep, d_num, B);
  hipLaunchKernelGGL(mv_keepbox_kernel, dim3(cdiv(keep_cap, 256), B), dim3(256), 0, s, d_order, d_keep, d_num, n, d_keepbox);
  MNC_HIP_TRY(hipGetLastError());
  std::vector<int> h_keepbox((size_t)B * n), h_num(B);
  MNC_HIP_TRY(hipMemcpyAsync(h_num.data(), d_num, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipMemcpyAsync(h_keepbox.data(), d_keepbox, (size_t)B * n * 4, hipMemcpyDeviceToHost, s));
  if (!on_device) MNC_HIP_TRY(hipMemcpyAsync((void*)d_masks, masks, (size_t)n * S * S * 4, hipMemcpyHostToDevice, s));
  MNC_HIP_TRY(hipStreamSynchronize(s));
  const auto t1 = now();

  // 2. global threshold = the max_per_image-th best kept score over all classes (:242-244)
  std::vector<float> pool;
  for (int c = 0; c < B; ++c)
    for (int k = 0; k < h_num[c]; ++k) pool.push_back(scores[(size_t)h_keepbox[(size_t)c * n + k] * num_classes + c + 1]);
  if (pool.empty()) { clear_error(); return MNC_OK; }
  std::vector<float> ranked(pool);
  const size_t kth = (ranked.size() < (size_t)max_per_image ? ranked.size() : (size_t)max_per_image) - 1;
  std::nth_element(ranked.begin(), ranked.begin() + kth, ranked.end(), std::greater<float>());
  const float thresh = ranked[kth];

  // 3. result rows = kept boxes at or above the threshold, class-major in keep order (:253-258); their candidate sets
  //    (:259-270) are built on the device
  std::vector<int> rows;
  int R = 0;
  for (int c = 0; c < B; ++c) {
    int cnt = 0;
    for (int k = 0; k < h_num[c]; ++k) {
      const int bi = h_keepbox[(size_t)c * n + k];
      const float sc = scores[(size_t)bi * num_classes + c + 1];
      if (!(sc >= thresh)) continue;
      rows.push_back(bi);
      rows.push_back(c);
      out_score[R++] = sc;
      ++cnt;
    }
    class_count[c] = cnt;
  }
  *result_num = R;
  if (R == 0) { clear_error(); return MNC_OK; }
  const auto t2 = now();

  // 4. candidate sets + the fused mask-voting kernels
  MNC_HIP_TRY(hipMemcpyAsync(d_rows, rows.data(), (size_t)R * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(mv_candidates_kernel, dim3(R), dim3(64), 0, s, d_boxes, d_scores, n, num_classes, d_rows, iou_thresh,
                     d_cinds, d_cw, d_cbegin, d_cend);
  mv_launch(s, d_boxes, 4, d_masks, S, d_cinds, d_cbegin, d_cend, d_cw, image_height, image_width, R, d_bounds, d_omask,
            d_obox);
  MNC_HIP_TRY(hipGetLastError());
  MNC_HIP_TRY(hipMemcpyAsync(out_mask, d_omask, (size_t)R * S * S * 4, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipMemcpyAsync(out_box, d_obox, (size_t)R * 16, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipStreamSynchronize(s));
  if (timing)
    fprintf(stderr, "mnc_mask_voting: order+nms+copies %.0f us, threshold+rows %.0f us (R=%d), candidates+voting+copies %.0f us\n",
            us(t0, t1), us(t1, t2), R, us(t2, now()));
  clear_error();
  return MNC_OK;
}

int mnc_mask_voting(const float* boxes, const float* masks, const float* scores, const int* order, int n, int num_classes,
                    int mask_size, int max_per_image, float nms_thresh, float iou_thresh, int image_height,
                    int image_width, float* out_mask, int* out_box, float* out_score, int* class_count, int* result_num,
                    int device_id) {
  return mask_voting_core(false, nullptr, boxes, masks, scores, order, n, num_classes, mask_size, max_per_image, nms_thresh,
                          iou_thresh, image_height, image_width, out_mask, out_box, out_score, class_count, result_num,
                          device_id);
}

int mnc_mask_voting_dev(mnc_ctx* ctx, const float* d_boxes, const float* d_masks, const float* d_scores, int n,
                        int num_classes, int mask_size, int max_per_image, float nms_thresh, float iou_thresh,
                        int image_height, int image_width, float* out_mask, int* out_box, float* out_score, int* class_count,
                        int* result_num) {
  MNC_REQUIRE(ctx, "mnc_mask_voting_dev: null context");
  MNC_REQUIRE(n == 0 || (d_boxes && d_masks && d_scores), "mnc_mask_voting_dev: null pointer");
  return mask_voting_core(true, ctx->stream, d_boxes, d_masks, d_scores, nullptr, n, num_classes, mask_size, max_per_image,
                          nms_thresh, iou_thresh, image_height, image_width, out_mask, out_box, out_score, class_count,
                          result_num, ctx->device);
}

// im_detect's tail on the device (tools/demo.py:84-100, TesterWrapper.py:240-260): boxes = clip(rois[:, 1:5] / scale) of both
// stages, stacked; float32 division and the clamp order of transform/bbox_transform.py:clip_boxes.
__global__ void detect_tail_kernel(const float* __restrict__ rois1, int R1, const float* __restrict__ rois2, int R2, float scale,
                                   float xmax, float ymax, float* __restrict__ boxes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (R1 + R2) * 4) return;
  const int r = i >> 2, k = i & 3;
  const float* roi = r < R1 ? rois1 + (long)r * 5 : rois2 + (long)(r - R1) * 5;
  const float v = roi[1 + k] / scale;
  const float hi = (k & 1) ? ymax : xmax;
  boxes[i] = fmaxf(fminf(v, hi), 0.0f);
}

int mnc_detect_tail(mnc_ctx* ctx, const float* d_rois1, int R1, const float* d_rois2, int R2, float scale, int image_height,
                    int image_width, float* d_boxes) {
  MNC_REQUIRE(ctx && d_boxes && R1 >= 0 && R2 >= 0 && (R1 == 0 || d_rois1) && (R2 == 0 || d_rois2) && scale > 0.0f,
              "mnc_detect_tail: bad argument");
  if (R1 + R2 == 0) return MNC_OK;
  LaunchScope ls(ctx, "detect_tail");
  hipLaunchKernelGGL(detect_tail_kernel, dim3(cdiv((R1 + R2) * 4, 256)), dim3(256), 0, ctx->stream, d_rois1, R1, d_rois2, R2,
                     scale, (float)(image_width - 1), (float)(image_height - 1), d_boxes);
  return ls.finish("detect_tail_kernel");
}

void _mv(const float* all_boxes, const float* all_masks, const int all_boxes_num, const int* candidate_inds,
         const int* candidate_start, const float* candidate_weights, const int candidate_num, const int image_height,
         const int image_width, const int box_dim, const int mask_size, const int result_num, float* out_mask,
         int* out_box, const int device_id) {
  if (mnc_mv(all_boxes, all_masks, all_boxes_num, candidate_inds, candidate_start, candidate_weights, candidate_num,
             image_height, image_width, box_dim, mask_size, result_num, out_mask, out_box, device_id) != MNC_OK)
    fprintf(stderr, "mnc_hip: _mv failed: %s\n", mnc_last_error());
}

}  // extern "C"
