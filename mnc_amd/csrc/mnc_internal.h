// Internal header of libmnc_hip.so (gfx950 only).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mnc_hip.h"

namespace mnc {

void set_error(const char* fmt, ...);
void clear_error();

struct ProfRecord {
  const char* name;
  hipEvent_t start, stop;
  double flops, bytes;
};

}  // namespace mnc

namespace mnc {
// Per-context overrides of the launchers' own choices (tile shapes, kernel variants, plan switches): mnc_ctx_set_tuning(ctx, "FC_TILE", 5)
// or, once, at context creation, the environment variable MNC_<NAME>.  They exist so that every variant a launcher can pick by
// shape is reachable from a test at a small shape, and for A/B measurements; no launch path calls getenv.  Ablation and
// superseded kernel builds (FC_ABL, FC_DMA_ABL, FCX3_ABL, CONV_ABL, WINO_V = 1, WINO_VAR != 7) are only compiled with -DMNC_TUNING.
#define MNC_TUNE_KEYS(X)                                                                                                          \
  X(CONV_COT) X(CONV_ROWS) X(CONV_KSPLIT) X(CONV_ABL) X(CONV1X1_TILE) X(CONV2D_WIDE) X(WINO_ROWS) X(WINO_TAIL) X(WINO_V) X(WINO_VAR)  \
  X(WINO_DMA) X(WINO_XCD) X(CONVX3_TILE) X(FC_NOTAIL) X(FC_TILE) X(FC_ABL) X(FC_DMA) X(PLAN) X(FC_RANGE_K) X(FC_SLOTS) X(FC_EVEN) X(FC_SPLIT_DIV) X(WINO_FILL) X(CONVX3_P0MIN) X(CONVX3_P1MIN) X(FC_DMA_ABL) X(FC_DMA_WAVES)    \
  X(FC_NO256) X(FCX3_TILE) X(FC_ORDER) X(FCX3_ABL) X(FC_SM) X(PACKED_ACT) X(FUSE_POOLS) X(BRANCH_STREAMS) X(TOPK_SINGLE_WG)           \
  X(ROI_SM_VARIANT) X(ROI_WARP_VARIANT) X(FC_REDUCE) X(WINO_F4) X(FUSE_SMALL) X(FCX3_WIDE) X(FC_HALF) X(WINO_STREAM) X(FC_MFMA16) X(WINO_MFMA16) X(ROI_ROW_SEGS)
enum TuneKey {
#define MNC_TUNE_ENUM(n) T_##n,
  MNC_TUNE_KEYS(MNC_TUNE_ENUM)
#undef MNC_TUNE_ENUM
  T_COUNT
};
constexpr int kTuneUnset = INT_MIN;
constexpr int kTickets = 4096;     // mnc_ctx::tickets
}  // namespace mnc

struct mnc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int profiling = 0;   // 0 off, 1 every launch, 2 only launches with >= 1 GFLOP of algorithmic work (the MFMA kernels)
  std::vector<mnc::ProfRecord> prof;
  std::vector<hipEvent_t> event_pool;
  // split-K scratch for mnc_fc and friends; grown on demand, never shrunk
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* proposal = nullptr;   // mnc_proposal_state (proposal.hip), created on first use
  void* vote_ws = nullptr;    // gpu_mask_voting scratch (mv.hip), grown on demand
  size_t vote_ws_bytes = 0;
  void* comm = nullptr;       // RCCL communicator state (comm.hip), set by mnc_comm_init
  // Arrival tickets of the K-range reductions that finish INSIDE the launch (gemm.hip, conv_wino4.hip): kTickets counters, zero
  // between launches -- allocated and zeroed with the context, every launch's last arriver of a tile puts its counter back to
  // zero.  Launches of one context are stream-ordered, so all of them share the array.
  unsigned* tickets = nullptr;
  // mnc_fc with defer_reduce set leaves the K ranges' partial sums in `scratch` ([splits][M][N]) instead of launching
  // fc_reduce_kernel and reports them here: the caller's next kernel sums them in range order itself (pipeline.hip: the sibling
  // classifiers' reduction, softmax, stage bridge and im_detect tail as one launch).  deferred_splits == 1: `out` is complete.
  bool defer_reduce = false;
  const float* deferred_part = nullptr;
  int deferred_splits = 0;
  // Bumped whenever one of the context-owned arenas above (scratch, proposal state, voting scratch) is re-allocated: a captured
  // HIP graph holds their raw addresses, so a graph owner (pipeline.hip) records the value at capture and drops its graph when
  // the value has moved on.
  unsigned long arena_gen = 0;
  // conventions of the three Caffe layers whose source is unavailable (all zero = oracle/SPEC.md); read by roi.hip's launchers
  mnc_layer_conventions conv = {0, 0, 0, 0, 0, 0, 0.4f, 0};
  bool capturing = false;     // between mnc_ctx_capture_begin / _end
  unsigned long capture_gen = 0;
  int tune[mnc::T_COUNT];     // kTuneUnset = the launcher decides (filled by mnc_ctx_create; mnc_ctx_set_tuning)
};

struct mnc_graph {
  hipGraphExec_t exec = nullptr;
  mnc_ctx* ctx = nullptr;
  unsigned long arena_gen = 0;
};

namespace mnc {
inline bool tune_set(const mnc_ctx* ctx, TuneKey k) { return ctx->tune[k] != kTuneUnset; }
inline int tune(const mnc_ctx* ctx, TuneKey k, int dflt) { return ctx->tune[k] != kTuneUnset ? ctx->tune[k] : dflt; }
// PLAN (round 6, profiles/r06_fc_ranges.txt): what the launchers' plans minimise.  0 (default) = the CU TIME of a launch -- the
// deployment the headline measures, several images in flight per GPU: the CUs a launch leaves free run the other images' kernels.
// 1 = the DURATION of a launch (rounds 1-5: every product cut until it fills the chip) -- one image at a time, latency.  One value
// per context (MNC_PLAN=1 or mnc_ctx_set_tuning(ctx, "PLAN", "1")); the nets that share results bit for bit must share it.
inline bool plan_latency(const mnc_ctx* ctx) { return tune(ctx, T_PLAN, 0) == 1; }
// K ranges of a reduced-precision InnerProduct over ONE row block (round 6, profiles/r06_fc_ranges.txt).  Rounds 2-5 cut K so that
// tiles x ranges filled all 256 CUs -- the shortest launch when the product has the chip to itself.  With several images in flight it
// does not: other images' kernels run on the CUs a launch leaves free, and what a product costs is its CU TIME.  Every range pays a
// prologue, an epilogue that writes 300 KB of partial sums, and its share of the reduction pass -- fc7 + fc7_mask in fp16: 17 us of
// MFMA loop inside 38 us + a 16 us reduction with 8 ranges of 8 stages.  So: ranges of at least 2048 K values (pairs: 12288, below), and
// no more ranges than fill HALF the chip (first: fc6 + fc6_mask 8 -> 4 ranges, fc7 + fc7_mask 8 -> 2; then 2 and 1).  Four images in flight: f16 870 -> 914 images/s, bf16
// 841 -> 902, mixed 553 -> 579, bf16x3 451 -> 472; one image at a time f16 566 -> 555, bf16x3 372 -> 322 (the price: a launch is
// longer).  One range for fc7 is the same throughput and 5-8 % more latency.  The fp32 InnerProducts keep the full cut (their loops
// are 10x longer than their fixed costs: 266 -> 260 images/s with half the ranges).
// FC_SPLIT_DIV (A/B): 0 = the full cut everywhere; otherwise the full cut divided by the low decimal digit (K > 8192) / the high
// digit (K <= 8192; 0 = the low digit), fp32 included.
inline int fc_split_div(const mnc_ctx* ctx, int splits, int K) {
  const int v = tune(ctx, T_FC_SPLIT_DIV, 1), lo = v % 10, hi = (v / 10) % 10 ? (v / 10) % 10 : lo, top = (v / 100) % 10 ? (v / 100) % 10 : lo;
  const int d = K <= 8192 ? hi : K > 50000 ? top : lo;      // (hundreds digit: K > 50000, fc6_maskest)
  return d > 1 && splits > 1 ? (splits / d > 1 ? splits / d : 1) : splits;
}
inline int fc_lowp_ranges(const mnc_ctx* ctx, int splits, int K, int tiles, bool pair = false) {
  if (tune_set(ctx, T_FC_SPLIT_DIV)) return fc_split_div(ctx, splits, K);
  if (plan_latency(ctx)) return splits;
  // pairs (fc6 + fc6_mask, fc7 + fc7_mask): ranges of >= 12288 K values -- 2 ranges for the fc6 pair, none for the fc7 pair (no
  // partial sums, no reduction launch): with the images in flight on 16 hardware queues (12 in flight) f16 1087 -> 1118 images/s,
  // mixed 652 -> 660, bf16x3 506 -> 513; with four in flight 1078 -> 1072 / 638 -> 645 / 499 -> 499; no cut at all (32768): 1119 /
  // 662 / 515 with twelve but 984 / 627 / 457 with four.  A single product (fc6_maskest: one column tile) keeps 2048.  FC_RANGE_K.
  const int per = pair ? tune(ctx, T_FC_RANGE_K, 12288) : 2048, part = 128;
  int r = K / per > 1 ? K / per : 1;
  const int half = (part + tiles - 1) / tiles;
  if (r > half) r = half;
  return r < splits ? r : splits;
}
}  // namespace mnc

namespace mnc {

int ensure_scratch(mnc_ctx* ctx, size_t bytes);
void prof_begin(mnc_ctx* ctx, const char* name, double flops, double bytes);
void prof_end(mnc_ctx* ctx);

#define MNC_HIP_TRY(expr)                                                                    \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      mnc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return MNC_ERR_HIP;                                                                    \
    }                                                                                        \
  } while (0)

// Entry points that synchronise the stream or re-allocate an arena refuse to run while a launch sequence is being captured on the
// context (mnc_ctx_capture_begin): the call fails BEFORE it touches the stream, so the capture itself stays valid and can be ended
// and discarded cleanly (a hipStreamSynchronize inside a capture leaves the stream unusable on this runtime).
#define MNC_NO_CAPTURE(ctx, what)                                                                        \
  do {                                                                                                   \
    if ((ctx)->capturing) {                                                                              \
      mnc::set_error("%s: synchronises or re-allocates; not allowed while a launch sequence is captured", what); \
      return MNC_ERR_STATE;                                                                              \
    }                                                                                                    \
  } while (0)

#define MNC_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      mnc::set_error(__VA_ARGS__);   \
      return MNC_ERR_INVALID;        \
    }                                \
  } while (0)

// RAII bracket: records a HIP event pair around the kernels launched inside its scope when profiling is on, and
// turns a failed launch into MNC_ERR_HIP.
struct LaunchScope {
  mnc_ctx* ctx;
  bool on;
  LaunchScope(mnc_ctx* c, const char* name, double flops = 0.0, double bytes = 0.0) : ctx(c) {
    (void)hipSetDevice(ctx->device);    // several contexts on different GPUs may live in one process
    on = ctx->profiling == 1 || (ctx->profiling == 2 && flops >= 1.0e9);
    if (on) prof_begin(ctx, name, flops, bytes);
  }
  int finish(const char* name) {
    if (on) prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("kernel launch %s failed: %s", name, hipGetErrorString(e));
      return MNC_ERR_HIP;
    }
    return MNC_OK;
  }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// K-split count of a GEMM whose `tiles` output tiles do not fill the chip on their own, by a small cost model instead of
// "tiles x splits ~ #CUs": workgroups run in rounds of `slots` (the CUs x workgroups per CU), a split of `per` stages costs
// per * stage_us, and every split adds a pass over the M x N partial sums (mn_bytes each way at ~3 TB/s) to the reduction.
// Used when there are several row blocks (M > one block: the CFM / ResNet configurations), where rounding the split count up
// could leave a second, nearly empty round (96 tiles x 3 splits = 288 workgroups on 256 CUs: 72 instead of 100 TFLOP/s).
static inline int choose_splits(int tiles, int stages, int min_stages, int slots, double stage_us, double mn_bytes) {
  int best = 1;
  double best_cost = 1e300;
  const int smax = stages / min_stages > 1 ? stages / min_stages : 1;
  for (int s = 1; s <= smax && s <= 1024; ++s) {
    const int per = cdiv(stages, s);
    if (cdiv(stages, per) != s) continue;                    // this count is not reachable after rounding `per` up
    const double rounds = (double)cdiv((long)tiles * s, slots);
    const double cost = rounds * per * stage_us + (s > 1 ? 3.0 + 2.0 * s * mn_bytes / 3.0e6 : 0.0);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

// XCD-aware block order for the GEMMs.  The dispatcher places block b on XCD b % 8 (observed, used for speed only:
// a different placement changes nothing but L2 hit rates).  Blocks are re-numbered so that each XCD receives a
// CONTIGUOUS range of the logical order (column tile fastest, then K split, then row block): the workgroups that share
// one K split's activation panel then sit on one XCD and the panel stays in that XCD's 4 MB L2 instead of being
// re-fetched from Infinity Cache by every column tile.  Bijective for any block count (cdna_hip_programming.md, T1).
__device__ __forceinline__ void xcd_decode(int bid, int tn, int splits, int tm, int& bn, int& split, int& bm) {
  const int total = tn * splits * tm;
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bn = logical % tn;
  const int rest = logical / tn;
  split = rest % splits;
  bm = rest / splits;
}

// ---- K ranges finished inside the launch (cdna_hip_programming.md section 5, "In-launch split-K reduction"; Guideline 16) ----
// Every workgroup of an output tile stores its partial sums as a SLAB in its own register layout (16-byte pieces, lane-contiguous,
// write-through `sc1` stores: no release fence), drains them, and draws an arrival ticket; the workgroup that draws the last one
// reads all slabs of the tile (sc1 loads behind one agent-scope acquire), adds them IN RANGE ORDER -- the order of the separate
// reduction kernels, so the results are the same bits whichever workgroup arrives last -- and writes the finished tile.  Nobody
// waits for anybody: a launch whose workgroups are not co-resident (several streams share the GPU) cannot deadlock.
// Memory-model assumption (MI355X_MICROARCH.md, "Valid forms"): sc1 stores are write-through to the memory side, the asm
// `s_waitcnt vmcnt(0)` in front of the barrier retires them before the ticket's relaxed agent-scope atomic can issue, and the last
// arriver reads them behind ONE agent-scope acquire -- there is no formal release fence; the ordering is the drained write-through.
// A launch that is aborted midway leaves tickets non-zero: mnc_ctx_sync zeroes the array when the stream reports an error.
typedef unsigned mnc_u32x4 __attribute__((ext_vector_type(4)));
typedef float mnc_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void slab_store(__amdgpu_buffer_rsrc_t rs, int voff, int soff, mnc_f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mnc_u32x4, v), rs, voff, soff, /*sc1*/ 16);
}
__device__ __forceinline__ mnc_f32x4 slab_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(mnc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, /*sc1*/ 16));
}
// All threads of the workgroup call it after their slab stores; `flag` is a free LDS word of the kernel's ONE shared array (a
// second __shared__ object de-pipelines LDS-DMA loops).  True in every thread of the tile's last arriver, which also puts the
// ticket back to zero for the next launch and has acquired the other workgroups' slabs.
__device__ __forceinline__ bool slab_last_arriver(unsigned* ticket, int arrivals, volatile unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through slab stores have left
  __syncthreads();
  if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const bool last = __builtin_amdgcn_readfirstlane(*flag) == (unsigned)(arrivals - 1);
  if (last) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  return last;
}

// Per-device stream + growable device buffer behind the host-pointer entry points (_nms/_mv): the reference
// cudaMalloc/cudaFree's its scratch on every call (nms_kernel.cu:99-143, mv_kernel.cu:250-347).
struct LegacyWs {
  std::mutex mu;
  hipStream_t stream = nullptr;
  void* buf = nullptr;
  size_t cap = 0;
};
// Returns the device's workspace with its mutex HELD in *lock (taken before the buffer may be re-allocated: ctypes releases
// the GIL, so two host threads may be inside _nms / _mv / mnc_mask_voting on one device) and at least `bytes` of buffer.
int legacy_ws(int device_id, size_t bytes, LegacyWs** out, std::unique_lock<std::mutex>* lock);  // nms.hip

// launchers shared between translation units (all asynchronous on `stream`, pointers are device pointers)
int nms_mask_launch(hipStream_t stream, const float* d_boxes, const int* d_order, int n, int dim, float thr,
                    unsigned long long* d_mask, int batch);
int nms_scan_launch(hipStream_t stream, const unsigned long long* d_mask, int n, int max_keep, int* d_keep, int* d_num,
                    int batch);
// same, with the box count read from device memory (*d_n <= n_cap); buffers are sized for n_cap
int nms_mask_launch_indirect(hipStream_t stream, const float* d_boxes, const int* d_order, const int* d_n, int n_cap,
                             int dim, float thr, unsigned long long* d_mask);
int nms_scan_launch_indirect(hipStream_t stream, const unsigned long long* d_mask, const int* d_n, int n_cap, int max_keep,
                             int* d_keep, int* d_num, const float* d_gather_boxes = nullptr, const int* d_gather_order = nullptr,
                             float* d_rois = nullptr, int rois_cap = 0);   // d_rois: the ProposalLayer's RoI rows written by the same launch
void proposal_state_free(void* state);  // proposal.hip
void comm_free(mnc_ctx* ctx);           // comm.hip
// out = act(sum of the ksplit partial c8 tensors + bias)  (conv.hip)
void fc_reduce_launch(hipStream_t stream, const float* part, const float* bias, float* out, int M, int N, int ldc, int splits,
                      int act);   // gemm.hip: out = act(sum of the K splits' partial sums + bias), shared by the three FC kernels
bool fc_reduce_launch_sm(hipStream_t stream, const float* part, const float* bias, float* out, int M, int N, int ldc, int splits,
                         int act, void* sm, int sm_fmt, long sm_rows, long sm_row0);   // + the rows in the next InnerProduct's form
bool fc_reduce_pair_launch_sm(hipStream_t stream, const float* part0, const float* part1, const float* bias0, const float* bias1,
                              float* out0, float* out1, int M, int N, int ldc, int splits, int act, void* sm0, void* sm1, int sm_fmt,
                              long sm_rows, bool* sm_done);   // two products' reductions in one launch (gemm.hip)
void conv_splitk_reduce_launch(hipStream_t stream, const float* d_part, const float* d_bias, float* d_out, int H, int W,
                               int Cout, int ksplit, int relu);
// proposal.hip: finishes the sibling classifiers of a head stage in one launch -- the K ranges of [cls_score | seg_cls_score |
// bbox_pred] summed in range order + bias (fc_reduce_kernel's arithmetic), the softmax of the seg_cls_score columns
// (softmax_rows_wave_kernel's), then StageBridgeLayer.forward_test (rois_ext != nullptr) or im_detect's tail (boxes != nullptr).
int heads_finish_launch(mnc_ctx* ctx, const float* part, int splits, const float* bias, float* heads, int ld, int M, int K,
                        float* scores, const float* rois, float im_h, float im_w, float* rois_ext, const float* rois1,
                        const float* rois2, float scale, int image_height, int image_width, float* boxes, const int* copy_src,
                        int* copy_dst);
// roi.hip: the pixel-major copy of a c8 feature map the warp kernels gather from, and the warp on a copy the caller already holds
int c8_to_hwc_launch(mnc_ctx* ctx, const float* d_feat, float* d_hwc, int C, int H, int W);
int roi_warp_from_hwc(mnc_ctx* ctx, const float* d_hwc, int C, int H, int W, const float* d_rois, int R, int PH, int PW, float scale,
                      int pool2, float* d_out, void* d_sm, int sm_fmt);
bool roi_warp_sm_only_ok(const mnc_ctx* ctx, int C, int pool2);      // may d_out be null (with a stage-major output)?
int detect_tail_launch(mnc_ctx* ctx, const float* d_rois1, int R1, const float* d_rois2, int R2, float scale, int image_height,
                       int image_width, float* d_boxes, const int* d_copy_src, int* d_copy_dst);   // mv.hip: mnc_detect_tail + one int moved
int mv_launch(hipStream_t stream, const float* d_boxes, int box_dim, const float* d_masks, int S, const int* d_inds,
              const int* d_begins, const int* d_ends, const float* d_wts, int H, int W, int R, int* d_bounds,
              float* d_out_mask, int* d_out_box);

}  // namespace mnc
