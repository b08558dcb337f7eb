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
// Result rows: r = blockIdx.x, blockIdx.x + gridDim.x, ... < R, where R = *rcount when the row count lives on the device (the
// asynchronous voting path picks the rows in mv_select_kernel and never tells the host) and the argument R otherwise.
__global__ __launch_bounds__(256) void mv_bounds_kernel(const float* __restrict__ boxes, int box_dim,
                                                        const float* __restrict__ masks, int S,
                                                        const int* __restrict__ inds, const int* __restrict__ begins,
                                                        const int* __restrict__ ends, const float* __restrict__ wts,
                                                        int H, int W, const int* __restrict__ rcount, int R,
                                                        int* __restrict__ bounds) {
  __shared__ CandLds cl;
  __shared__ int red[4][4];
  __shared__ int ubox[4];
  const int Rv = rcount ? *rcount : R;
  for (int r = blockIdx.x; r < Rv; r += gridDim.x) {
    __syncthreads();                      // the previous row's LDS contents are dead from here on
    const int c0 = begins[r], c1 = ends[r];
    const int nc = c1 - c0;
    if (nc <= 0) continue;
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
    if (ux2 < ux1 || uy2 < uy1) continue;
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
}

// block 448 (7 waves; 441 active), rows as in mv_bounds_kernel.  Finalises the box (defaults W/2, H/2, :149,:173) and
// resamples (:193-240).
__global__ __launch_bounds__(448) void mv_resample_kernel(const float* __restrict__ boxes, int box_dim,
                                                          const float* __restrict__ masks, int S,
                                                          const int* __restrict__ inds, const int* __restrict__ begins,
                                                          const int* __restrict__ ends, const float* __restrict__ wts,
                                                          int H, int W, const int* __restrict__ bounds,
                                                          const int* __restrict__ rcount, int R,
                                                          float* __restrict__ out_mask, int* __restrict__ out_box,
                                                          float* __restrict__ records = nullptr, const float* __restrict__ rscore = nullptr,
                                                          const int* __restrict__ rows = nullptr, int record_cap = 0) {
  __shared__ CandLds cl;
  const int Rv = rcount ? *rcount : R;
  // Round 6: the fixed-shape records (x1, y1, x2, y2, score, class id 1..B, S*S mask values; rows from the count up to record_cap
  // zero: class 0 == padding -- SURVEY.md 8e) are written here as well, what mv_pack_kernel did as a launch of its own
  const int D = 6 + S * S;
  if (records)
    for (int r = Rv + blockIdx.x; r < record_cap; r += gridDim.x)
      for (int i = threadIdx.x; i < D; i += blockDim.x) records[(long)r * D + i] = 0.0f;
  for (int r = blockIdx.x; r < Rv; r += gridDim.x) {
    __syncthreads();
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
    float* rec = (records && r < record_cap) ? records + (long)r * D : nullptr;
    if (rec && threadIdx.x == 64) {                  // (another wave than lane 0's stores above)
      rec[0] = (float)bx1; rec[1] = (float)by1; rec[2] = (float)bx2; rec[3] = (float)by2;
      rec[4] = rscore[r];
      rec[5] = (float)(rows[2 * r + 1] + 1);
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
      if (rec) rec[6 + idx] = v;
    }
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

constexpr int kSelThreads = 1024;
constexpr int kSelCap = 8192;        // kept boxes over all classes the in-LDS selection sorts (20 classes x 100: 2000)
constexpr int kSelMaxClasses = 1024;

// The host part of gpu_mask_voting between the per-class NMS and the candidate sets (mask_transform.py:242-258), on the device,
// ONE workgroup:  pool = kept scores of all classes (class-major, keep order; num[c] <= max_per_image already);
//   thresh = np.sort(pool)[::-1][min(len(pool), max_per_image) - 1]      (NaN sorts FIRST in that order, as numpy's does)
//   rows   = [(box, class) for class-major kept boxes with score >= thresh]            (NaN >= x is false)
// Outputs: rows [R][2] = (box index, class - 1), rscore [R], counts[0] = R, counts[c] = rows of class c (1..B).
// pool_* are global scratch of P entries.  NP = power of two >= P capacity (dynamic LDS: NP keys).
// (round 5: the kept boxes are read through order[keep[]] here -- the mv_keepbox launch that materialised them is gone)
__global__ __launch_bounds__(kSelThreads) void mv_select_kernel(const float* __restrict__ scores, int n, int C,
                                                                const int* __restrict__ order, const int* __restrict__ keep,
                                                                const int* __restrict__ num,
                                                                int max_per_image, int NP, int* __restrict__ pool_box,
                                                                float* __restrict__ pool_score, int* __restrict__ pool_cls,
                                                                int* __restrict__ rows, float* __restrict__ rscore,
                                                                int* __restrict__ counts) {
  extern __shared__ unsigned s_key[];
  __shared__ int s_off[kSelMaxClasses + 1];
  __shared__ int s_cc[kSelMaxClasses];
  __shared__ int s_wsum[kSelThreads / 64];
  const int tid = threadIdx.x, B = C - 1;
  for (int c = tid; c < B; c += kSelThreads) s_cc[c] = 0;
  if (tid == 0) {
    int acc = 0;
    for (int c = 0; c < B; ++c) { s_off[c] = acc; acc += num[c]; }
    s_off[B] = acc;
  }
  __syncthreads();
  const int P = s_off[B];
  if (P == 0) {
    for (int c = tid; c < C; c += kSelThreads) counts[c] = 0;
    return;
  }
  for (int j = tid; j < NP; j += kSelThreads) {
    unsigned key = 0xFFFFFFFFu;                     // padding sorts last
    if (j < P) {
      int lo = 0, hi = B - 1;                       // class c with s_off[c] <= j < s_off[c + 1] (empty classes skipped)
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_off[mid] <= j) lo = mid; else hi = mid - 1;
      }
      const int c = lo, k = j - s_off[c];
      const int bi = order[(long)c * n + keep[(long)c * n + k]];   // keep lists index the sorted order; the voting wants box indices
      const float v = scores[(long)bi * C + c + 1];
      pool_box[j] = bi; pool_cls[j] = c; pool_score[j] = v;
      if (v != v) {
        key = 0u;                                   // NaN first
      } else {
        const unsigned u = __float_as_uint(v);
        const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // ascending-orderable; >= 0x007FFFFF
        key = ~asc;                                 // descending; +inf -> 0x007FFFFF > 0, -inf -> 0xFF7FFFFF < padding
      }
    }
    s_key[j] = key;
  }
  __syncthreads();
  for (int size = 2; size <= NP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (NP >> 1); t += kSelThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned x = s_key[lo], y = s_key[hi];
        if ((x > y) == up) { s_key[lo] = y; s_key[hi] = x; }
      }
      __syncthreads();
    }
  }
  const unsigned kkey = s_key[min(P, max_per_image) - 1];
  float thresh;
  if (kkey == 0u) {
    thresh = __uint_as_float(0x7FC00000u);
  } else {
    const unsigned asc = ~kkey;
    thresh = __uint_as_float((asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc);
  }
  // rows in pool order: each thread owns a contiguous run, block-wide exclusive scan of the run counts
  const int per = (P + kSelThreads - 1) / kSelThreads;
  const int j0 = min(tid * per, P), j1 = min(j0 + per, P);
  int mine = 0;
  for (int j = j0; j < j1; ++j) mine += (pool_score[j] >= thresh) ? 1 : 0;
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if ((tid & 63) >= o) inc += t;
  }
  if ((tid & 63) == 63) s_wsum[tid >> 6] = inc;
  __syncthreads();
  int base = inc - mine;
  for (int w = 0; w < (tid >> 6); ++w) base += s_wsum[w];
  for (int j = j0; j < j1; ++j) {
    const float v = pool_score[j];
    if (v >= thresh) {
      rows[2 * base] = pool_box[j];
      rows[2 * base + 1] = pool_cls[j];
      rscore[base] = v;
      atomicAdd(&s_cc[pool_cls[j]], 1);
      ++base;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int total = 0;
    for (int w = 0; w < kSelThreads / 64; ++w) total += s_wsum[w];
    counts[0] = total;
  }
  for (int c = tid; c < B; c += kSelThreads) counts[c + 1] = s_cc[c];
}

// Candidate set of result row r = (kept box bi, class c): members {i : IoU_f64(box_i, box_bi) >= iou_thresh} in index order,
// weights = class scores / sum (mask_transform.py:259-267).  The reference's `cur_weights / sum(cur_weights)` is python's
// sum() over float32 scalars: under the numpy 1.x the reference ran on, 0 + np.float32 promotes to float64 (scalar-scalar
// promotion of a python int), so the accumulation is sequential in float64, and the float32 array is then divided by that
// float64 scalar in float32 (value-based casting): w / float32(sum64).  One wave per row; row r owns cinds/cw[r*n .. r*n + n).
__global__ __launch_bounds__(64) void mv_candidates_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           int n, int C, const int* __restrict__ rows,
                                                           const int* __restrict__ rcount, int R, float iou_thresh,
                                                           int* __restrict__ cinds, float* __restrict__ cw,
                                                           int* __restrict__ cbegin, int* __restrict__ cend,
                                                           int* __restrict__ bounds) {
  const int lane = threadIdx.x;
  const int Rv = rcount ? *rcount : R;
  for (int r = blockIdx.x; r < Rv; r += gridDim.x) {
    if (bounds && lane < 4) bounds[r * 4 + lane] = lane < 2 ? INT_MAX : -1;      // (mv_init_bounds_kernel's values, for mv_bounds_kernel)
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
    double sum = 0.0;
    if (lane == 0)
      for (int t = 0; t < base; ++t) sum += (double)my_w[t];              // ((0 + w0) + w1) + ... in float64
    const float sum32 = __shfl((float)sum, 0);
    __syncthreads();
    for (int t = lane; t < base; t += 64) my_w[t] = my_w[t] / sum32;
    if (lane == 0) {
      cbegin[r] = r * n;
      cend[r] = r * n + base;
    }
    __syncthreads();
  }
}

// All pointers device.  d_bounds: R*4 ints scratch.  Result r's candidates are d_inds/d_wts[d_begins[r] .. d_ends[r]).
// d_rcount != nullptr: the row count is read from the device (<= R rows of scratch/output exist) and the grid is `grid_rows`
// workgroups striding over the rows.
static int mv_launch_impl(hipStream_t stream, const float* d_boxes, int box_dim, const float* d_masks, int S, const int* d_inds,
                          const int* d_begins, const int* d_ends, const float* d_wts, int H, int W, int R, const int* d_rcount,
                          int grid_rows, int* d_bounds, float* d_out_mask, int* d_out_box, bool bounds_ready = false,
                          float* d_records = nullptr, const float* d_rscore = nullptr, const int* d_rows = nullptr, int record_cap = 0) {
  if (R <= 0) return MNC_OK;
  // (the voting sequence initialises the rows' bounds in its candidates kernel; the extension's own entry point does it here)
  if (!bounds_ready) hipLaunchKernelGGL(mv_init_bounds_kernel, dim3(cdiv(R * 4, 256)), dim3(256), 0, stream, d_bounds, R);
  // ~2048 blocks in flight: rows x `splits` row slabs each
  int splits = 2048 / grid_rows;
  if (splits < 1) splits = 1;
  if (splits > 32) splits = 32;
  hipLaunchKernelGGL(mv_bounds_kernel, dim3(grid_rows, splits), dim3(256), 0, stream, d_boxes, box_dim, d_masks, S, d_inds,
                     d_begins, d_ends, d_wts, H, W, d_rcount, R, d_bounds);
  hipLaunchKernelGGL(mv_resample_kernel, dim3(grid_rows), dim3(448), 0, stream, d_boxes, box_dim, d_masks, S, d_inds, d_begins,
                     d_ends, d_wts, H, W, d_bounds, d_rcount, R, d_out_mask, d_out_box, d_records, d_rscore, d_rows, record_cap);
  return MNC_OK;
}

int mv_launch(hipStream_t stream, const float* d_boxes, int box_dim, const float* d_masks, int S, const int* d_inds,
              const int* d_begins, const int* d_ends, const float* d_wts, int H, int W, int R, int* d_bounds,
              float* d_out_mask, int* d_out_box) {
  return mv_launch_impl(stream, d_boxes, box_dim, d_masks, S, d_inds, d_begins, d_ends, d_wts, H, W, R, nullptr, R, d_bounds,
                        d_out_mask, d_out_box);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Device scratch of one gpu_mask_voting problem (n boxes, C classes incl. background, S x S masks, keep_cap kept per class).
struct VoteWs {
  float *boxes, *masks, *scores;        // staging for host inputs (unused when the inputs are already on the device)
  int* order; unsigned long long* bits; int *keep, *num;
  int *pool_box, *pool_cls; float* pool_score;
  int* rows; float* rscore; int* counts;
  int *cinds; float* cw; int *cbegin, *cend, *bounds; float* omask; int* obox;
  float* records;                       // [Rmax][6 + S*S]
};

static size_t vote_ws_layout(char* base, int n, int C, int S, int keep_cap, VoteWs* w) {
  const int B = C - 1, cb = cdiv(n, 64), Rmax = B * keep_cap;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base + off; off += align256(bytes); return p; };
  w->boxes = (float*)take((size_t)n * 16);
  w->masks = (float*)take((size_t)n * S * S * 4);
  w->scores = (float*)take((size_t)n * C * 4);
  w->order = (int*)take((size_t)B * n * 4);
  w->bits = (unsigned long long*)take((size_t)B * n * cb * 8);
  w->keep = (int*)take((size_t)B * n * 4);
  w->num = (int*)take((size_t)B * 4);
  w->pool_box = (int*)take((size_t)Rmax * 4);
  w->pool_cls = (int*)take((size_t)Rmax * 4);
  w->pool_score = (float*)take((size_t)Rmax * 4);
  w->rows = (int*)take((size_t)Rmax * 8);
  w->rscore = (float*)take((size_t)Rmax * 4);
  w->counts = (int*)take((size_t)C * 4);
  w->cinds = (int*)take((size_t)Rmax * n * 4);
  w->cw = (float*)take((size_t)Rmax * n * 4);
  w->cbegin = (int*)take((size_t)Rmax * 4);
  w->cend = (int*)take((size_t)Rmax * 4);
  w->bounds = (int*)take((size_t)Rmax * 16);
  w->omask = (float*)take((size_t)Rmax * S * S * 4);
  w->obox = (int*)take((size_t)Rmax * 16);
  w->records = (float*)take((size_t)Rmax * (6 + S * S) * 4);
  return off;
}

// gpu_mask_voting (lib/transform/mask_transform.py:213-286 + lib/nms/mv_kernel.cu) as ONE asynchronous launch sequence on
// `s`, inputs and outputs on the device, no host decision anywhere:
//   per-class order (unless order_ready) -> 20 batched NMS problems -> kept box indices -> threshold + result rows
//   (mv_select_kernel) -> candidate sets -> fused voting kernels -> records.
// d_records [record_cap][6+S*S] (record_cap <= B*keep_cap), d_counts [C].
static int vote_async(hipStream_t s, const VoteWs& w, const float* d_boxes, const float* d_masks, const float* d_scores,
                      bool order_ready, int n, int C, int S, int max_per_image, float nms_thresh, float iou_thresh, int H, int W,
                      float* d_records, int record_cap, int* d_counts) {
  const int B = C - 1;
  const int keep_cap = max_per_image < n ? max_per_image : n;
  const int Rmax = B * keep_cap;
  if (!order_ready) {
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(mv_order_kernel, dim3(B), dim3(np2 < 1024 ? np2 : 1024), (size_t)np2 * 8, s, d_scores, n, C, np2, w.order);
  }
  nms_mask_launch(s, d_boxes, w.order, n, 4, nms_thresh, w.bits, B);
  nms_scan_launch(s, w.bits, n, keep_cap, w.keep, w.num, B);
  int NP = 64;
  while (NP < Rmax) NP <<= 1;
  hipLaunchKernelGGL(mv_select_kernel, dim3(1), dim3(kSelThreads), (size_t)NP * 4, s, d_scores, n, C, w.order, w.keep, w.num,
                     max_per_image, NP, w.pool_box, w.pool_score, w.pool_cls, w.rows, w.rscore, d_counts);
  // the row count R = d_counts[0] stays on the device: R <= max_per_image unless scores tie at the threshold, so that many
  // workgroups stride over the rows
  const int grid_rows = Rmax < max_per_image ? Rmax : max_per_image;
  hipLaunchKernelGGL(mv_candidates_kernel, dim3(grid_rows), dim3(64), 0, s, d_boxes, d_scores, n, C, w.rows, d_counts, Rmax,
                     iou_thresh, w.cinds, w.cw, w.cbegin, w.cend, w.bounds);
  // (round 6: the resampling kernel writes the records too -- mv_pack_kernel's launch is gone)
  mv_launch_impl(s, d_boxes, 4, d_masks, S, w.cinds, w.cbegin, w.cend, w.cw, H, W, Rmax, d_counts, grid_rows, w.bounds, w.omask,
                 w.obox, /*bounds_ready=*/true, d_records, w.rscore, w.rows, record_cap);
  MNC_HIP_TRY(hipGetLastError());
  return MNC_OK;
}

static int vote_check_args(const char* who, int n, int C, int S, int max_per_image, int H, int W) {
  MNC_REQUIRE(n >= 0 && C >= 2 && S >= 2 && max_per_image > 0 && H > 0 && W > 0, "%s: bad argument", who);
  MNC_REQUIRE(C - 1 <= kSelMaxClasses, "%s: %d classes exceed the device selection's %d", who, C - 1, kSelMaxClasses);
  const long Rmax = (long)(C - 1) * (max_per_image < n ? max_per_image : n);
  MNC_REQUIRE(Rmax <= kSelCap, "%s: (num_classes-1) * min(max_per_image, n) = %ld exceeds the device selection's %d kept boxes",
              who, Rmax, kSelCap);
  return MNC_OK;
}

// Per-class descending order on the host for box counts beyond the LDS sort (the same order: stable, NaN last).
static void host_order(const float* scores, int n, int C, std::vector<int>* out) {
  const int B = C - 1;
  out->resize((size_t)B * n);
  for (int c = 0; c < B; ++c) {
    int* o = out->data() + (size_t)c * n;
    for (int i = 0; i < n; ++i) o[i] = i;
    const float* sc = scores + c + 1;
    std::stable_sort(o, o + n, [&](int a, int b) {
      const float va = -sc[(size_t)a * C], vb = -sc[(size_t)b * C];
      return va < vb || (vb != vb && va == va);              // NaN last, as numpy
    });
  }
}

// Host-facing outputs of a finished vote: counts + records from the device, split into the reference's arrays.
static int vote_fetch(hipStream_t s, const float* d_records, const int* d_counts, int Rmax, int C, int S, int first_rows,
                      float* out_mask, int* out_box, float* out_score, int* class_count, int* result_num) {
  const int D = 6 + S * S;
  std::vector<int> counts(C);
  std::vector<float> rec;
  int first = first_rows < Rmax ? first_rows : Rmax;
  rec.resize((size_t)first * D);
  MNC_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)C * 4, hipMemcpyDeviceToHost, s));
  if (first) MNC_HIP_TRY(hipMemcpyAsync(rec.data(), d_records, rec.size() * 4, hipMemcpyDeviceToHost, s));
  MNC_HIP_TRY(hipStreamSynchronize(s));
  const int R = counts[0];
  if (R > first) {                              // scores tied at the threshold: more rows than max_per_image
    rec.resize((size_t)R * D);
    MNC_HIP_TRY(hipMemcpyAsync(rec.data() + (size_t)first * D, d_records + (size_t)first * D, (size_t)(R - first) * D * 4,
                               hipMemcpyDeviceToHost, s));
    MNC_HIP_TRY(hipStreamSynchronize(s));
  }
  for (int r = 0; r < R; ++r) {
    const float* q = rec.data() + (size_t)r * D;
    for (int k = 0; k < 4; ++k) out_box[r * 4 + k] = (int)q[k];
    out_score[r] = q[4];
    memcpy(out_mask + (size_t)r * S * S, q + 6, (size_t)S * S * 4);
  }
  for (int c = 0; c < C - 1; ++c) class_count[c] = counts[c + 1];
  *result_num = R;
  return MNC_OK;
}

}  // namespace mnc

using namespace mnc;

// per-context voting scratch (mnc_ctx::vote_ws), grown on demand
static int ctx_vote_ws(mnc_ctx* ctx, int n, int C, int S, int keep_cap, VoteWs* w) {
  const size_t need = vote_ws_layout(nullptr, n, C, S, keep_cap, w);
  if (need > ctx->vote_ws_bytes) {
    MNC_NO_CAPTURE(ctx, "voting scratch growth");
    MNC_HIP_TRY(hipSetDevice(ctx->device));
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->vote_ws) MNC_HIP_TRY(hipFree(ctx->vote_ws));
    ctx->vote_ws = nullptr;
    ctx->vote_ws_bytes = 0;
    ++ctx->arena_gen;                  // a captured graph that holds the old address must not be replayed (pipeline.hip)
    hipError_t e = hipMalloc(&ctx->vote_ws, need + (need >> 2));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("voting scratch: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
      return MNC_ERR_NOMEM;
    }
    ctx->vote_ws_bytes = need + (need >> 2);
  }
  vote_ws_layout((char*)ctx->vote_ws, n, C, S, keep_cap, w);
  return MNC_OK;
}

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
  std::unique_lock<std::mutex> lock;
  int rc = legacy_ws(device_id, b_boxes + b_masks + b_inds + b_wts + b_starts + b_bounds + b_omask + b_obox, &w, &lock);
  if (rc) return rc;
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

// gpu_mask_voting in one call on HOST arrays (see include/mnc_hip.h): inputs up, the asynchronous device sequence, records down.
int mnc_mask_voting(const float* boxes, const float* masks, const float* scores, const int* order, int n, int num_classes,
                    int mask_size, int max_per_image, float nms_thresh, float iou_thresh, int image_height,
                    int image_width, float* out_mask, int* out_box, float* out_score, int* class_count, int* result_num,
                    int device_id) {
  MNC_REQUIRE(result_num && class_count, "mnc_mask_voting: null output pointer");
  int rc = vote_check_args("mnc_mask_voting", n, num_classes, mask_size, max_per_image, image_height, image_width);
  if (rc) return rc;
  const int B = num_classes - 1, S = mask_size, C = num_classes;
  *result_num = 0;
  for (int c = 0; c < B; ++c) class_count[c] = 0;
  if (n == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(boxes && masks && scores && out_mask && out_box && out_score, "mnc_mask_voting: null pointer");
  if (order)
    for (long i = 0; i < (long)B * n; ++i)
      MNC_REQUIRE(order[i] >= 0 && order[i] < n, "mnc_mask_voting: order[%ld]=%d out of range", i, order[i]);
  const int keep_cap = max_per_image < n ? max_per_image : n;
  VoteWs ws;
  const size_t need = vote_ws_layout(nullptr, n, C, S, keep_cap, &ws);
  LegacyWs* w = nullptr;
  std::unique_lock<std::mutex> lock;
  rc = legacy_ws(device_id, need, &w, &lock);
  if (rc) return rc;
  vote_ws_layout((char*)w->buf, n, C, S, keep_cap, &ws);
  hipStream_t s = w->stream;
  MNC_HIP_TRY(hipMemcpyAsync(ws.boxes, boxes, (size_t)n * 16, hipMemcpyHostToDevice, s));
  MNC_HIP_TRY(hipMemcpyAsync(ws.scores, scores, (size_t)n * C * 4, hipMemcpyHostToDevice, s));
  MNC_HIP_TRY(hipMemcpyAsync(ws.masks, masks, (size_t)n * S * S * 4, hipMemcpyHostToDevice, s));
  std::vector<int> h_order;
  if (!order && n > kMaxOrderDevice) {
    host_order(scores, n, C, &h_order);
    order = h_order.data();
  }
  if (order) MNC_HIP_TRY(hipMemcpyAsync(ws.order, order, (size_t)B * n * 4, hipMemcpyHostToDevice, s));
  rc = vote_async(s, ws, ws.boxes, ws.masks, ws.scores, order != nullptr, n, C, S, max_per_image, nms_thresh, iou_thresh,
                  image_height, image_width, ws.records, B * keep_cap, ws.counts);
  if (rc) return rc;
  rc = vote_fetch(s, ws.records, ws.counts, B * keep_cap, C, S, max_per_image, out_mask, out_box, out_score, class_count,
                  result_num);
  if (rc) return rc;
  clear_error();
  return MNC_OK;
}

// The same with inputs AND outputs on the device, asynchronous on the context's stream -- what the whole-image path uses.
int mnc_vote_instances(mnc_ctx* ctx, const float* d_boxes, const float* d_masks, const float* d_scores, int n, int num_classes,
                       int mask_size, int max_per_image, float nms_thresh, float iou_thresh, int image_height, int image_width,
                       float* d_records, int record_cap, int* d_counts) {
  MNC_REQUIRE(ctx && d_records && d_counts && record_cap >= 0, "mnc_vote_instances: null pointer");
  int rc = vote_check_args("mnc_vote_instances", n, num_classes, mask_size, max_per_image, image_height, image_width);
  if (rc) return rc;
  const int B = num_classes - 1, S = mask_size, C = num_classes;
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  if (n == 0) {
    MNC_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)C * 4, ctx->stream));
    if (record_cap) MNC_HIP_TRY(hipMemsetAsync(d_records, 0, (size_t)record_cap * (6 + S * S) * 4, ctx->stream));
    return MNC_OK;
  }
  MNC_REQUIRE(d_boxes && d_masks && d_scores, "mnc_vote_instances: null pointer");
  MNC_REQUIRE(n <= kMaxOrderDevice, "mnc_vote_instances: n=%d exceeds the device ordering's %d boxes", n, kMaxOrderDevice);
  const int keep_cap = max_per_image < n ? max_per_image : n;
  VoteWs ws;
  rc = ctx_vote_ws(ctx, n, C, S, keep_cap, &ws);
  if (rc) return rc;
  const int Rmax = B * keep_cap;
  LaunchScope ls(ctx, "mask_voting");
  if (record_cap > Rmax) {
    MNC_HIP_TRY(hipMemsetAsync(d_records + (size_t)Rmax * (6 + S * S), 0, (size_t)(record_cap - Rmax) * (6 + S * S) * 4,
                               ctx->stream));
    record_cap = Rmax;
  }
  rc = vote_async(ctx->stream, ws, d_boxes, d_masks, d_scores, false, n, C, S, max_per_image, nms_thresh, iou_thresh,
                  image_height, image_width, d_records, record_cap, d_counts);
  if (rc) return rc;
  return ls.finish("mask_voting");
}

int mnc_mask_voting_dev(mnc_ctx* ctx, const float* d_boxes, const float* d_masks, const float* d_scores, int n,
                        int num_classes, int mask_size, int max_per_image, float nms_thresh, float iou_thresh,
                        int image_height, int image_width, float* out_mask, int* out_box, float* out_score, int* class_count,
                        int* result_num) {
  MNC_REQUIRE(ctx && result_num && class_count, "mnc_mask_voting_dev: null pointer");
  int rc = vote_check_args("mnc_mask_voting_dev", n, num_classes, mask_size, max_per_image, image_height, image_width);
  if (rc) return rc;
  const int B = num_classes - 1, S = mask_size, C = num_classes;
  *result_num = 0;
  for (int c = 0; c < B; ++c) class_count[c] = 0;
  if (n == 0) { clear_error(); return MNC_OK; }
  MNC_REQUIRE(d_boxes && d_masks && d_scores && out_mask && out_box && out_score, "mnc_mask_voting_dev: null pointer");
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  const int keep_cap = max_per_image < n ? max_per_image : n;
  VoteWs ws;
  rc = ctx_vote_ws(ctx, n, C, S, keep_cap, &ws);
  if (rc) return rc;
  bool order_ready = false;
  if (n > kMaxOrderDevice) {                  // beyond the LDS sort: the same order on the host (one extra round trip)
    std::vector<float> h_scores((size_t)n * C);
    std::vector<int> h_order;
    MNC_HIP_TRY(hipMemcpyAsync(h_scores.data(), d_scores, h_scores.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    host_order(h_scores.data(), n, C, &h_order);
    MNC_HIP_TRY(hipMemcpyAsync(ws.order, h_order.data(), h_order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    order_ready = true;
  }
  rc = vote_async(ctx->stream, ws, d_boxes, d_masks, d_scores, order_ready, n, C, S, max_per_image, nms_thresh, iou_thresh,
                  image_height, image_width, ws.records, B * keep_cap, ws.counts);
  if (rc) return rc;
  rc = vote_fetch(ctx->stream, ws.records, ws.counts, B * keep_cap, C, S, max_per_image, out_mask, out_box, out_score,
                  class_count, result_num);
  if (rc) return rc;
  clear_error();
  return MNC_OK;
}

// im_detect's tail on the device (tools/demo.py:84-100, TesterWrapper.py:240-260): boxes = clip(rois[:, 1:5] / scale) of both
// stages, stacked; float32 division and the clamp order of transform/bbox_transform.py:clip_boxes.
// copy_src / copy_dst (optional): one int carried along -- the whole-image pipeline moves the ProposalLayer's row count into the
// header of its result block here, so that counts, proposal count and records come down in ONE copy (pipeline.hip).
__global__ void detect_tail_kernel(const float* __restrict__ rois1, int R1, const float* __restrict__ rois2, int R2, float scale,
                                   float xmax, float ymax, float* __restrict__ boxes, const int* __restrict__ copy_src,
                                   int* __restrict__ copy_dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && copy_src) *copy_dst = *copy_src;
  if (i >= (R1 + R2) * 4) return;
  const int r = i >> 2, k = i & 3;
  const float* roi = r < R1 ? rois1 + (long)r * 5 : rois2 + (long)(r - R1) * 5;
  const float v = roi[1 + k] / scale;
  const float hi = (k & 1) ? ymax : xmax;
  boxes[i] = fmaxf(fminf(v, hi), 0.0f);
}

}  // extern "C"
namespace mnc {
int detect_tail_launch(mnc_ctx* ctx, const float* d_rois1, int R1, const float* d_rois2, int R2, float scale, int image_height,
                       int image_width, float* d_boxes, const int* d_copy_src, int* d_copy_dst) {
  MNC_REQUIRE(ctx && d_boxes && R1 >= 0 && R2 >= 0 && (R1 == 0 || d_rois1) && (R2 == 0 || d_rois2) && scale > 0.0f,
              "mnc_detect_tail: bad argument");
  if (R1 + R2 == 0 && !d_copy_src) return MNC_OK;
  LaunchScope ls(ctx, "detect_tail");
  const int blocks = cdiv((R1 + R2) * 4, 256);
  hipLaunchKernelGGL(detect_tail_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, ctx->stream, d_rois1, R1, d_rois2, R2,
                     scale, (float)(image_width - 1), (float)(image_height - 1), d_boxes, d_copy_src, d_copy_dst);
  return ls.finish("detect_tail_kernel");
}
}  // namespace mnc
extern "C" {

int mnc_detect_tail(mnc_ctx* ctx, const float* d_rois1, int R1, const float* d_rois2, int R2, float scale, int image_height,
                    int image_width, float* d_boxes) {
  return mnc::detect_tail_launch(ctx, d_rois1, R1, d_rois2, R2, scale, image_height, image_width, d_boxes, nullptr, nullptr);
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
