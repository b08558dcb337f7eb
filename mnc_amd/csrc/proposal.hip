// Device-resident forms of MNC's three inference-time Python layers for gfx950:
//   ProposalLayer.forward      (lib/pylayer/proposal_layer.py:52-175)   -> mnc_proposal
//   MaskLayer.forward_test     (lib/pylayer/mask_layer.py:95-102)       -> a device copy (done by the engine)
//   StageBridgeLayer.forward_test (lib/pylayer/stage_bridge_layer.py:237-255) -> mnc_stage_bridge
// The reference runs them on the host inside net.forward (4 blob round trips per image); the Python classes stay the
// API (mnc_amd/lib/pylayer) and remain the path for user-defined layers, these kernels are what the engine substitutes
// for the three stock classes.
//
// Proposal pipeline, all on the engine's stream, no host round trip until the final 4-byte count:
//   1. decode   one thread per anchor (h, w, a): fg score = channel A+a, box = bbox_transform_inv(anchor, deltas) in the
//               reference's float32 operation order, clip to the image, min-size test.          (proposal_layer.py:75-132)
//   2. top-k    ONE workgroup: 8-pass radix select on the 64-bit key (descending score, ascending anchor index) finds
//               the pre_nms_topN-th key, survivors are compacted into LDS and bitonic-sorted there.        (:139-145)
//               Tie order is therefore DEFINED (score desc, index asc) where the reference's `argsort()[::-1]` leaves it
//               to numpy's unstable sort; for distinct scores the orders coincide.
//   3. NMS      the bitmask + single-wave scan kernels of nms.hip, reading the boxes through the sorted index list and
//               the candidate count from device memory, stopping at post_nms_topN survivors.                (:149-153)
//   4. gather   rois[r] = (0, box[order[keep[r]]]).                                                           (:159-160)
// exp: np_exp.h restates numpy's float32 exp loop bit for bit, and the decode is written in bbox_transform_inv's float32
// operation order (-ffp-contract=off), so the decoded boxes -- and with them rois / rois_ext -- carry the same bits as the
// reference's numpy Python layers on the same input blobs (tests/test_np_exp.py on the host, tests/test_gpu_engine.py on the GPU).
#include <cfloat>
#include <cstdlib>

#include <atomic>

#include "mnc_internal.h"
#include "np_exp.h"

namespace mnc {

typedef unsigned long long u64;

struct Anchors { float v[16][4]; int count; };

__device__ __forceinline__ unsigned desc_key(float s) {
  // ascending order of the result == descending order of s
  unsigned b = __float_as_uint(s);
  unsigned asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ~asc;
}

// boxes [N][4]; keys [N] = (desc_key(score) << 32 | index), or all-ones for filtered boxes; scores_out [N]
__global__ __launch_bounds__(256) void proposal_decode_kernel(const float* __restrict__ prob, const float* __restrict__ deltas,
                                                              Anchors anc, int A, int H, int W, float stride, float im_h,
                                                              float im_w, float min_size, float* __restrict__ boxes,
                                                              u64* __restrict__ keys, float* __restrict__ scores) {
  const int N = H * W * A;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int a = i % A, cell = i / A;
  const int w = cell % W, h = cell / W;
  const long hw = (long)H * W, off = (long)h * W + w;
  const float score = prob[(long)(A + a) * hw + off];
  const float dx = deltas[(long)(4 * a + 0) * hw + off], dy = deltas[(long)(4 * a + 1) * hw + off];
  const float dw = deltas[(long)(4 * a + 2) * hw + off], dh = deltas[(long)(4 * a + 3) * hw + off];
  const float sx = (float)w * stride, sy = (float)h * stride;
  const float ax1 = anc.v[a][0] + sx, ay1 = anc.v[a][1] + sy, ax2 = anc.v[a][2] + sx, ay2 = anc.v[a][3] + sy;
  // bbox_transform_inv, lib/transform/bbox_transform.py:74-97
  const float widths = ax2 - ax1 + 1.0f, heights = ay2 - ay1 + 1.0f;
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = np_exp_f32(dw) * widths, ph = np_exp_f32(dh) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  // clip_boxes, :110-119
  x1 = fmaxf(fminf(x1, im_w - 1.0f), 0.0f); y1 = fmaxf(fminf(y1, im_h - 1.0f), 0.0f);
  x2 = fmaxf(fminf(x2, im_w - 1.0f), 0.0f); y2 = fmaxf(fminf(y2, im_h - 1.0f), 0.0f);
  // filter_small_boxes, :123-130
  const bool ok = (x2 - x1 + 1.0f >= min_size) && (y2 - y1 + 1.0f >= min_size);
  reinterpret_cast<float4*>(boxes)[i] = make_float4(x1, y1, x2, y2);
  scores[i] = score;
  keys[i] = ok ? (((u64)desc_key(score) << 32) | (unsigned)i) : ~0ull;
}

constexpr int kSelThreads = 1024;
constexpr int kSortCap = 16384;      // pre_nms_topN up to 16384 (TEST 6000, TRAIN 12000)
constexpr int kKeysPer = 24;         // candidate keys held in registers per thread: 24 576 anchors (600x1000: 21 546)

// ONE workgroup.  Selects the K smallest keys (K = min(topn, #valid)), sorts them ascending in LDS and writes
// order[j] = anchor index, sorted_scores[j]; *n_out = K.
__global__ __launch_bounds__(kSelThreads) void proposal_topk_kernel(const u64* __restrict__ keys, const float* __restrict__ scores,
                                                                    int N, int topn, int* __restrict__ order,
                                                                    float* __restrict__ sorted_scores, int* __restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* sk = reinterpret_cast<u64*>(smem);                       // [cap] keys being sorted
  __shared__ unsigned hist[256];
  __shared__ unsigned wsum[4];
  __shared__ unsigned s_cnt;
  __shared__ u64 s_prefix;
  __shared__ unsigned s_remaining;
  __shared__ unsigned s_valid;
  const int tid = threadIdx.x;

  // The candidate keys are read ONCE into registers (kKeysPer per thread) and every pass below runs on them: with a single
  // workgroup the ten passes over global memory were latency-bound (21 dependent-looking L2 round trips per thread and
  // pass, ~150 us for 21 546 anchors).  Slots past N hold the "filtered" key ~0.  (N > kSelThreads * kKeysPer falls back to
  // re-reading global memory for the surplus.)
  u64 rk[kKeysPer];
#pragma unroll
  for (int u = 0; u < kKeysPer; ++u) {
    const int i = tid + u * kSelThreads;
    rk[u] = i < N ? keys[i] : ~0ull;
  }
  const int Nreg = min(N, kSelThreads * kKeysPer);

  // number of valid (unfiltered) boxes
  if (tid == 0) s_valid = 0;
  __syncthreads();
  unsigned local = 0;
#pragma unroll
  for (int u = 0; u < kKeysPer; ++u) local += rk[u] != ~0ull;
  for (int i = Nreg + tid; i < N; i += kSelThreads) local += keys[i] != ~0ull;
  atomicAdd(&s_valid, local);
  __syncthreads();
  const unsigned K = min((unsigned)topn, s_valid);
  if (K == 0) {
    if (tid == 0) *n_out = 0;
    return;
  }
  // radix select, most significant byte first: after pass p every key with (key >> shift) < prefix is selected and
  // the K-th smallest has (key >> shift) == prefix
  if (tid == 0) { s_prefix = 0; s_remaining = K; }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const u64 prefix = s_prefix;
    const unsigned rem = s_remaining;
#pragma unroll
    for (int u = 0; u < kKeysPer; ++u) {
      const u64 k = rk[u];
      if (k != ~0ull && (pass == 0 || (k >> (shift + 8)) == prefix)) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
    }
    for (int i = Nreg + tid; i < N; i += kSelThreads) {
      const u64 k = keys[i];
      if (k != ~0ull && (pass == 0 || (k >> (shift + 8)) == prefix)) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    // the bin holding the rem-th smallest remaining key: parallel inclusive scan of the 256 counts (a serial walk by one
    // thread is 256 dependent LDS reads -- ~7 us per pass)
    unsigned v = 0, inc = 0;
    if (tid < 256) {
      v = hist[tid];
      inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if ((tid & 63) >= o) inc += t;
      }
      if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      unsigned off = 0;
      for (int w = 0; w < (tid >> 6); ++w) off += wsum[w];
      inc += off;
      const unsigned exc = inc - v;
      if (exc < rem && rem <= inc) {                     // exactly one bin satisfies this (rem >= 1, counts sum >= rem)
        s_remaining = rem - exc;
        s_prefix = (prefix << 8) | (unsigned)tid;
      }
    }
    __syncthreads();
  }
  const u64 kth = s_prefix;                                     // the K-th smallest key (keys are unique)
  // compact the K survivors into LDS, pad to a power of two with +inf, bitonic sort ascending
  unsigned cap = 1;
  while (cap < K) cap <<= 1;
  if (tid == 0) s_cnt = 0;
  for (unsigned j = tid; j < cap; j += kSelThreads) sk[j] = ~0ull;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kKeysPer; ++u)
    if (rk[u] <= kth) sk[atomicAdd(&s_cnt, 1u)] = rk[u];
  for (int i = Nreg + tid; i < N; i += kSelThreads) {
    const u64 k = keys[i];
    if (k <= kth) sk[atomicAdd(&s_cnt, 1u)] = k;
  }
  __syncthreads();
  for (unsigned size = 2; size <= cap; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned t = tid; t < (cap >> 1); t += kSelThreads) {
        const unsigned lo = 2 * t - (t & (stride - 1));
        const unsigned hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const u64 a = sk[lo], b = sk[hi];
        if ((a > b) == up) { sk[lo] = b; sk[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (unsigned j = tid; j < K; j += kSelThreads) {
    const int idx = (int)(unsigned)(sk[j] & 0xFFFFFFFFull);
    order[j] = idx;
    sorted_scores[j] = scores[idx];
  }
  if (tid == 0) *n_out = (int)K;
}

// ---- multi-workgroup top-K: sort runs + rank merge -------------------------------------------------------------------
// The single-workgroup kernel above is a ~100 us latency chain at 600x1000 (one CU busy, 255 idle).  Same result from two
// wide launches:
//   1. proposal_sort_runs_kernel   one workgroup per run of kRun keys: bitonic sort in LDS, sorted run back to global memory
//                                  (filtered keys ~0 sort to the end of their run);
//   2. proposal_rank_kernel        one thread per key: its rank in the whole set = position in its own run + for every other
//                                  run the number of smaller keys (lower_bound by binary search; the searches of one thread
//                                  are independent chains, so their L2 latencies overlap).  Keys are unique (the anchor
//                                  index is part of the key), so ranks are a permutation: key of rank r < K goes to slot r.
// The output (order, sorted_scores, n_out) is identical to proposal_topk_kernel's.
constexpr int kRun = 2048;
static_assert(kRun == 1 << 11, "proposal_rank_kernel's step count is written for 2^11-element runs");
constexpr int kMaxRuns = 32;          // up to 65 536 anchors (1000x1000: 35 721); larger maps use the single-workgroup kernel

__global__ __launch_bounds__(1024) void proposal_sort_runs_kernel(const u64* __restrict__ keys, int N, u64* __restrict__ runs) {
  __shared__ u64 sk[kRun];
  const int base = blockIdx.x * kRun;
  for (int i = threadIdx.x; i < kRun; i += 1024) sk[i] = base + i < N ? keys[base + i] : ~0ull;
  __syncthreads();
  for (unsigned size = 2; size <= kRun; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned t = threadIdx.x;                       // kRun / 2 == 1024 compare-exchanges per step
      const unsigned lo = 2 * t - (t & (stride - 1));
      const unsigned hi = lo + stride;
      const bool up = ((lo & size) == 0);
      const u64 a = sk[lo], b = sk[hi];
      if ((a > b) == up) { sk[lo] = b; sk[hi] = a; }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < kRun; i += 1024) runs[base + i] = sk[i];
}

__global__ __launch_bounds__(256) void proposal_rank_kernel(const u64* __restrict__ runs, const float* __restrict__ scores,
                                                            int nruns, int topn, int* __restrict__ order,
                                                            float* __restrict__ sorted_scores, int* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // position in the concatenated runs
  const int total = nruns * kRun;
  const u64 key = i < total ? runs[i] : ~0ull;
  const int mine = i / kRun;
  // lower_bound(key) in every run at once: the interval [lo[r], hi[r]) halves per step; kRun = 2^11 elements need 12 steps to
  // reach the empty interval (2048 -> 1024 -> ... -> 1 -> 0).
  int lo[kMaxRuns], hi[kMaxRuns];
#pragma unroll
  for (int r = 0; r < kMaxRuns; ++r) { lo[r] = 0; hi[r] = kRun; }
#pragma unroll 1
  for (int step = 0; step < 12; ++step) {
#pragma unroll
    for (int r = 0; r < kMaxRuns; ++r) {
      if (r < nruns && lo[r] < hi[r]) {
        const int mid = (lo[r] + hi[r]) >> 1;
        const u64 v = runs[r * kRun + mid];
        if (v < key) lo[r] = mid + 1; else hi[r] = mid;
      }
    }
  }
  int rank = 0;
#pragma unroll
  for (int r = 0; r < kMaxRuns; ++r)
    if (r < nruns) rank += (r == mine) ? (i - mine * kRun) : lo[r];
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    // number of valid (unfiltered) keys = sum over runs of lower_bound(~0) (runs are sorted, sentinels at the end): lane r of
    // the grid's first wave searches run r, then a wave reduction
    int a = 0;
    if ((int)threadIdx.x < nruns) {
      int b = kRun;
      const u64* run = runs + (long)threadIdx.x * kRun;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (run[mid] != ~0ull) a = mid + 1; else b = mid;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (threadIdx.x == 0) *n_out = min(topn, a);
  }
  if (key != ~0ull && rank < topn) {
    const int idx = (int)(unsigned)(key & 0xFFFFFFFFull);
    order[rank] = idx;
    sorted_scores[rank] = scores[idx];
  }
}

// StageBridgeLayer.forward_test: box of argmax class (first maximum, background included), decoded and clipped.
__global__ void stage_bridge_kernel(const float* __restrict__ rois, const float* __restrict__ bbox_pred, int ld_bbox,
                                    const float* __restrict__ probs, int ld_probs, int R, int K, float im_h, float im_w,
                                    float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* p = probs + (long)r * ld_probs;
  int best = 0;
  float bv = p[0];
  for (int k = 1; k < K; ++k)
    if (p[k] > bv) { bv = p[k]; best = k; }
  const float x1 = rois[r * 5 + 1], y1 = rois[r * 5 + 2], x2 = rois[r * 5 + 3], y2 = rois[r * 5 + 4];
  const float* d = bbox_pred + (long)r * ld_bbox + 4 * best;
  const float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;
  const float ctr_x = x1 + 0.5f * widths, ctr_y = y1 + 0.5f * heights;
  const float pcx = d[0] * widths + ctr_x, pcy = d[1] * heights + ctr_y;
  const float pw = np_exp_f32(d[2]) * widths, ph = np_exp_f32(d[3]) * heights;
  float ox1 = pcx - 0.5f * pw, oy1 = pcy - 0.5f * ph, ox2 = pcx + 0.5f * pw, oy2 = pcy + 0.5f * ph;
  ox1 = fmaxf(fminf(ox1, im_w - 1.0f), 0.0f); oy1 = fmaxf(fminf(oy1, im_h - 1.0f), 0.0f);
  ox2 = fmaxf(fminf(ox2, im_w - 1.0f), 0.0f); oy2 = fmaxf(fminf(oy2, im_h - 1.0f), 0.0f);
  out[r * 5 + 0] = 0.f;
  out[r * 5 + 1] = ox1; out[r * 5 + 2] = oy1; out[r * 5 + 3] = ox2; out[r * 5 + 4] = oy2;
}

// One launch for what follows the sibling classifiers' GEMM of a head stage (test.prototxt:713-805 resp. :1040-1106 and
// tools/demo.py:84-100), one 128-thread block per RoI row m:
//   1. heads[m][n] = sum over the K ranges (in range order, from zero) + bias[n]   -- fc_reduce_kernel<1, 0>'s arithmetic (no activation);
//      splits == 1: the GEMM wrote the row itself;
//   2. scores[m][:] = softmax(heads[m][K .. 2K))                                    -- softmax_rows_wave_kernel's arithmetic (seg_cls_prob);
//   3. rois_ext[m] = StageBridgeLayer.forward_test on row m (stage_bridge_kernel's arithmetic)           when rois_ext != nullptr,
//      boxes[m], boxes[M + m] = im_detect's tail on rois1[m], rois2[m] (detect_tail_kernel's arithmetic)  when boxes != nullptr.
// The same bits as the four separate launches (tests/test_gpu_pipeline.py: the Python engine runs those).
__global__ __launch_bounds__(128) void heads_finish_kernel(const float* __restrict__ part, int splits, const float* __restrict__ bias,
                                                           float* __restrict__ heads, int ld, int M, int K, float* __restrict__ scores,
                                                           const float* __restrict__ rois, float im_h, float im_w,
                                                           float* __restrict__ rois_ext, const float* __restrict__ rois1,
                                                           const float* __restrict__ rois2, float scale, float xmax, float ymax,
                                                           float* __restrict__ boxes, const int* __restrict__ copy_src,
                                                           int* __restrict__ copy_dst) {
  __shared__ float row[6 * 64 + 64];                               // the row's 6K values, then its K probabilities
  const int m = blockIdx.x, t = threadIdx.x, N = 6 * K;
  if (m == 0 && t == 0 && copy_src) *copy_dst = *copy_src;
  for (int n = t; n < N; n += 128) {
    float v;
    if (splits > 1) {
      v = 0.f;
      const long total = (long)M * N, idx = (long)m * N + n;
      for (int s = 0; s < splits; ++s) v += part[(long)s * total + idx];
      v = v + bias[n];
      heads[(long)m * ld + n] = v;
    } else {
      v = heads[(long)m * ld + n];
    }
    row[n] = v;
  }
  __syncthreads();
  if (t >= 64) return;
  const int lane = t;
  const bool live = lane < K;
  const float v = live ? row[K + lane] : -INFINITY;
  float mx = v;
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float e = live ? expf(v - mx) : 0.f;
  float sum = 0.f;
  for (int i = 0; i < K; ++i) sum += __shfl(e, i);
  const float pr = e / sum;
  if (live) { scores[(long)m * K + lane] = pr; row[N + lane] = pr; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // lane 0 reads what lanes 0 .. K - 1 just wrote (one wave: no barrier left)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (rois_ext && lane == 0) {
    const float* p = row + N;
    int best = 0;
    float bv = p[0];
    for (int k = 1; k < K; ++k)
      if (p[k] > bv) { bv = p[k]; best = k; }
    const float x1 = rois[m * 5 + 1], y1 = rois[m * 5 + 2], x2 = rois[m * 5 + 3], y2 = rois[m * 5 + 4];
    const float* d = row + 2 * K + 4 * best;
    const float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;
    const float ctr_x = x1 + 0.5f * widths, ctr_y = y1 + 0.5f * heights;
    const float pcx = d[0] * widths + ctr_x, pcy = d[1] * heights + ctr_y;
    const float pw = np_exp_f32(d[2]) * widths, ph = np_exp_f32(d[3]) * heights;
    float ox1 = pcx - 0.5f * pw, oy1 = pcy - 0.5f * ph, ox2 = pcx + 0.5f * pw, oy2 = pcy + 0.5f * ph;
    ox1 = fmaxf(fminf(ox1, im_w - 1.0f), 0.0f); oy1 = fmaxf(fminf(oy1, im_h - 1.0f), 0.0f);
    ox2 = fmaxf(fminf(ox2, im_w - 1.0f), 0.0f); oy2 = fmaxf(fminf(oy2, im_h - 1.0f), 0.0f);
    rois_ext[m * 5 + 0] = 0.f;
    rois_ext[m * 5 + 1] = ox1; rois_ext[m * 5 + 2] = oy1; rois_ext[m * 5 + 3] = ox2; rois_ext[m * 5 + 4] = oy2;
  }
  if (boxes && lane < 8) {
    const int which = lane >> 2, k = lane & 3;
    const float* roi = (which ? rois2 : rois1) + (long)m * 5;
    const float q = roi[1 + k] / scale;
    const float hi = (k & 1) ? ymax : xmax;
    boxes[((long)which * M + m) * 4 + k] = fmaxf(fminf(q, hi), 0.0f);
  }
}

struct ProposalWs {
  float* boxes = nullptr; u64* keys = nullptr; float* scores = nullptr; u64* runs = nullptr;
  int* order = nullptr; float* sorted_scores = nullptr; int* n_cand = nullptr;
  u64* mask = nullptr; int* keep = nullptr; int* num = nullptr;
  int cap_n = 0, cap_k = 0;
};

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace mnc

using namespace mnc;

struct mnc_proposal_state {
  void* buf = nullptr;
  size_t bytes = 0;
  ProposalWs ws;
  int last_n = 0, last_topn = 0;
};

namespace mnc {
int* proposal_count_ptr(mnc_ctx* ctx) {
  mnc_proposal_state* st = (mnc_proposal_state*)ctx->proposal;
  return st ? st->ws.num : nullptr;
}
void proposal_state_free(void* state) {
  mnc_proposal_state* st = (mnc_proposal_state*)state;
  if (st->buf) (void)hipFree(st->buf);
  delete st;
}
}  // namespace mnc

namespace mnc {
int heads_finish_launch(mnc_ctx* ctx, const float* part, int splits, const float* bias, float* heads, int ld, int M, int K,
                        float* scores, const float* rois, float im_h, float im_w, float* rois_ext, const float* rois1,
                        const float* rois2, float scale, int image_height, int image_width, float* boxes, const int* copy_src,
                        int* copy_dst) {
  MNC_REQUIRE(ctx && bias && heads && scores && M > 0 && K >= 2 && K <= 64 && ld >= 6 * K && splits >= 1 && (splits == 1 || part),
              "heads_finish: bad argument");
  LaunchScope ls(ctx, "heads_finish");
  hipLaunchKernelGGL(heads_finish_kernel, dim3(M), dim3(128), 0, ctx->stream, part, splits, bias, heads, ld, M, K, scores, rois,
                     im_h, im_w, rois_ext, rois1, rois2, scale, (float)(image_width - 1), (float)(image_height - 1), boxes,
                     copy_src, copy_dst);
  return ls.finish("heads_finish_kernel");
}
}  // namespace mnc

static mnc_proposal_state* state_of(mnc_ctx* ctx) {
  if (!ctx->proposal) ctx->proposal = new mnc_proposal_state();
  return (mnc_proposal_state*)ctx->proposal;
}

extern "C" {

int mnc_proposal(mnc_ctx* ctx, const float* d_cls_prob, const float* d_bbox_pred, int A, int H, int W,
                 const float* anchors_host, int feat_stride, float im_h, float im_w, float im_scale, int pre_nms_topn,
                 int post_nms_topn, float nms_thresh, float min_size, float* d_rois, int* num_rois_host) {
  MNC_REQUIRE(ctx && d_cls_prob && d_bbox_pred && anchors_host && d_rois, "mnc_proposal: null pointer");
  MNC_REQUIRE(A > 0 && A <= 16 && H > 0 && W > 0 && post_nms_topn > 0, "mnc_proposal: bad shape A=%d H=%d W=%d", A, H, W);
  const int N = H * W * A;
  int topn = pre_nms_topn > 0 ? pre_nms_topn : N;
  if (topn > N) topn = N;
  MNC_REQUIRE(topn <= kSortCap, "mnc_proposal: pre_nms_topN=%d exceeds the LDS sort capacity %d", topn, kSortCap);
  const int cb = cdiv(topn, 64);
  mnc_proposal_state* st = state_of(ctx);
  const int nruns = cdiv(N, kRun);
  const bool wide_topk = nruns <= kMaxRuns && !tune(ctx, T_TOPK_SINGLE_WG, 0);
  const size_t need = a256((size_t)N * 16) + a256((size_t)N * 8) + a256((size_t)N * 4) + a256((size_t)topn * 4) * 2 + 256 +
                      a256((size_t)topn * cb * 8) + a256((size_t)topn * 4) + 256 + a256((size_t)nruns * kRun * 8);
  if (need > st->bytes) {
    MNC_NO_CAPTURE(ctx, "proposal state growth");
    MNC_HIP_TRY(hipSetDevice(ctx->device));
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (st->buf) MNC_HIP_TRY(hipFree(st->buf));
    st->buf = nullptr; st->bytes = 0;
    ++ctx->arena_gen;                  // a captured graph that holds the old address must not be replayed (pipeline.hip)
    hipError_t e = hipMalloc(&st->buf, need + (need >> 2));
    if (e != hipSuccess) { set_error("mnc_proposal: hipMalloc(%zu) failed", need); return MNC_ERR_NOMEM; }
    st->bytes = need + (need >> 2);
  }
  char* p = (char*)st->buf;
  ProposalWs& w = st->ws;
  w.boxes = (float*)p; p += a256((size_t)N * 16);
  w.keys = (u64*)p; p += a256((size_t)N * 8);
  w.scores = (float*)p; p += a256((size_t)N * 4);
  w.order = (int*)p; p += a256((size_t)topn * 4);
  w.sorted_scores = (float*)p; p += a256((size_t)topn * 4);
  w.n_cand = (int*)p; p += 256;
  w.mask = (u64*)p; p += a256((size_t)topn * cb * 8);
  w.keep = (int*)p; p += a256((size_t)topn * 4);
  w.num = (int*)p; p += 256;
  w.runs = (u64*)p;
  st->last_n = N; st->last_topn = topn;

  Anchors anc;
  anc.count = A;
  for (int a = 0; a < A; ++a)
    for (int k = 0; k < 4; ++k) anc.v[a][k] = anchors_host[a * 4 + k];
  {
    LaunchScope ls(ctx, "proposal_decode");
    hipLaunchKernelGGL(proposal_decode_kernel, dim3(cdiv(N, 256)), dim3(256), 0, ctx->stream, d_cls_prob, d_bbox_pred, anc, A,
                       H, W, (float)feat_stride, im_h, im_w, min_size * im_scale, w.boxes, w.keys, w.scores);
    int rc = ls.finish("proposal_decode_kernel");
    if (rc) return rc;
  }
  if (wide_topk) {
    LaunchScope ls(ctx, "proposal_topk");
    hipLaunchKernelGGL(proposal_sort_runs_kernel, dim3(nruns), dim3(1024), 0, ctx->stream, w.keys, N, w.runs);
    hipLaunchKernelGGL(proposal_rank_kernel, dim3(cdiv(nruns * kRun, 256)), dim3(256), 0, ctx->stream, w.runs, w.scores, nruns,
                       topn, w.order, w.sorted_scores, w.n_cand);
    int rc = ls.finish("proposal_rank_kernel");
    if (rc) return rc;
  } else {
    unsigned cap = 1;
    while (cap < (unsigned)topn) cap <<= 1;
    static std::atomic<unsigned long long> attr_set{0};      // one bit per device: function attributes are per device
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {   // up to 128 KB of dynamic LDS for the in-LDS sort
      MNC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_topk_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kSortCap * 8));
      attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    LaunchScope ls(ctx, "proposal_topk");
    hipLaunchKernelGGL(proposal_topk_kernel, dim3(1), dim3(kSelThreads), (size_t)cap * 8, ctx->stream, w.keys, w.scores, N,
                       topn, w.order, w.sorted_scores, w.n_cand);
    int rc = ls.finish("proposal_topk_kernel");
    if (rc) return rc;
  }
  {
    LaunchScope ls(ctx, "proposal_nms");
    nms_mask_launch_indirect(ctx->stream, w.boxes, w.order, w.n_cand, topn, 4, nms_thresh, w.mask);
    // (round 6: the scan's wave also writes the RoI rows -- the survivors' boxes through `order` -- one launch fewer per image)
    nms_scan_launch_indirect(ctx->stream, w.mask, w.n_cand, topn, post_nms_topn, w.keep, w.num, w.boxes, w.order, d_rois, post_nms_topn);
    int rc = ls.finish("proposal_nms");
    if (rc) return rc;
  }
  if (num_rois_host && ctx->capturing) {
    set_error("mnc_proposal: reading the row count back synchronises; pass NULL while a launch sequence is captured");
    return MNC_ERR_STATE;
  }
  if (num_rois_host) {                  // NULL: the count stays on the device (mnc_proposal_count), no synchronisation
    MNC_HIP_TRY(hipMemcpyAsync(num_rois_host, w.num, 4, hipMemcpyDeviceToHost, ctx->stream));
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  clear_error();
  return MNC_OK;
}

int mnc_proposal_count(mnc_ctx* ctx, int* num_rois_host) {
  MNC_REQUIRE(ctx && num_rois_host, "mnc_proposal_count: null pointer");
  mnc_proposal_state* st = (mnc_proposal_state*)ctx->proposal;
  MNC_REQUIRE(st && st->buf, "mnc_proposal_count: mnc_proposal has not run on this context");
  MNC_NO_CAPTURE(ctx, "mnc_proposal_count");
  MNC_HIP_TRY(hipMemcpyAsync(num_rois_host, st->ws.num, 4, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  clear_error();
  return MNC_OK;
}

int mnc_proposal_count_ptr(mnc_ctx* ctx, void** d_count) {
  MNC_REQUIRE(ctx && d_count, "mnc_proposal_count_ptr: null pointer");
  mnc_proposal_state* st = (mnc_proposal_state*)ctx->proposal;
  MNC_REQUIRE(st && st->buf, "mnc_proposal_count_ptr: mnc_proposal has not run on this context");
  *d_count = st->ws.num;
  clear_error();
  return MNC_OK;
}

int mnc_proposal_candidates(mnc_ctx* ctx, float* boxes_host, float* scores_host, int capacity, int* n_host) {
  MNC_REQUIRE(ctx && n_host, "mnc_proposal_candidates: null pointer");
  mnc_proposal_state* st = (mnc_proposal_state*)ctx->proposal;
  MNC_REQUIRE(st && st->buf, "mnc_proposal_candidates: mnc_proposal has not run on this context");
  MNC_NO_CAPTURE(ctx, "mnc_proposal_candidates");
  int n = 0;
  MNC_HIP_TRY(hipMemcpyAsync(&n, st->ws.n_cand, 4, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *n_host = n;
  if (n == 0 || !boxes_host || !scores_host) return MNC_OK;
  MNC_REQUIRE(capacity >= n, "mnc_proposal_candidates: capacity %d < %d candidates", capacity, n);
  std::vector<int> order(n);
  std::vector<float> all((size_t)st->last_n * 4);
  MNC_HIP_TRY(hipMemcpyAsync(order.data(), st->ws.order, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipMemcpyAsync(all.data(), st->ws.boxes, all.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipMemcpyAsync(scores_host, st->ws.sorted_scores, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < n; ++j) memcpy(boxes_host + (size_t)j * 4, all.data() + (size_t)order[j] * 4, 16);
  return MNC_OK;
}

int mnc_stage_bridge(mnc_ctx* ctx, const float* d_rois, const float* d_bbox_pred, int ld_bbox, const float* d_probs,
                     int ld_probs, int R, int K, float im_h, float im_w, float* d_rois_ext) {
  MNC_REQUIRE(ctx && R >= 0 && K > 0 && ld_bbox >= 4 * K && ld_probs >= K, "mnc_stage_bridge: bad argument");
  if (R == 0) return MNC_OK;
  MNC_REQUIRE(d_rois && d_bbox_pred && d_probs && d_rois_ext, "mnc_stage_bridge: null pointer");
  LaunchScope ls(ctx, "stage_bridge");
  hipLaunchKernelGGL(stage_bridge_kernel, dim3(cdiv(R, 64)), dim3(64), 0, ctx->stream, d_rois, d_bbox_pred, ld_bbox, d_probs,
                     ld_probs, R, K, im_h, im_w, d_rois_ext);
  return ls.finish("stage_bridge_kernel");
}

}  // extern "C"
