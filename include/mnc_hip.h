/* mnc_hip.h -- C ABI of libmnc_hip.so: MNC's per-image inference hot path on MI355X (gfx950).
 *
 * This header is the drop-in boundary.  Plain C: pointers, ints, floats; no torch / C++
 * types.  Every entry point returns an int status (MNC_OK == 0) except the two reference-compatible `void`
 * wrappers `_nms` / `_mv`; the text of the last error on the calling thread is available from mnc_last_error().
 * Citations `file:line` are into the reference repository (daijifeng001/MNC).
 *
 * Pointer naming:  *_host = host memory owned by the caller;  d_* = device memory on the context's GPU.
 *
 * Device tensor layouts ("c8" = channel-blocked, chosen so that a wave's MFMA epilogue stores and the next layer's
 * halo loads are both fully coalesced -- see DESIGN.md section 3):
 *   feature map        c8   float [C/8][H][W][8]                      (batch is always 1, proposal_layer.py:65)
 *   conv3x3 weights    packed by mnc_pack_conv3x3_weights             float [Cin/8][Cout][76]  (9 taps x 8 cin + 4 pad)
 *   per-RoI features   hwc  float [R][PH][PW][C]   (a row of the FC GEMM is one RoI, k = (ph*PW+pw)*C + c)
 *   FC weights         packed by mnc_pack_fc_weights: rows = outputs, columns permuted from Caffe's (c,h,w) order
 *                      (test.prototxt InnerProduct, SURVEY App. A graph-5) to the hwc order above
 *   everything else    as in Caffe: rois [R][5], masks [R][1][21][21], probabilities [R][21], ... row-major.
 */
#ifndef MNC_HIP_H_
#define MNC_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNC_API __attribute__((visibility("default")))

enum {
  MNC_OK = 0,
  MNC_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, unsupported shape) */
  MNC_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed */
  MNC_ERR_NOMEM = 3,       /* device or host allocation failed */
  MNC_ERR_STATE = 4,       /* call order violated (e.g. forward before weights were loaded) */
  MNC_ERR_UNSUPPORTED = 5  /* valid request this build cannot serve */
};

/* Thread-local, never NULL; "" when the last call on this thread succeeded. */
MNC_API const char* mnc_last_error(void);
MNC_API int mnc_device_count(int* count);
/* Free and total bytes of device `device_id` (hipMemGetInfo): what a host that keeps several nets per GPU checks its budget with
 * (bench.py prints total - free per rank: four images in flight x 1.1 GB of replicated weights + activations). */
MNC_API int mnc_device_mem_info(int device_id, size_t* free_bytes, size_t* total_bytes);
MNC_API const char* mnc_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * b1  nms.gpu_nms  --  replaces `_nms` (lib/nms/gpu_nms.hpp:1-2, lib/nms/nms_kernel.cu:91-144), called from
 *     gpu_nms.pyx:16-31.  boxes_host: [boxes_num][boxes_dim] float32 ALREADY SORTED by descending score; only
 *     columns 0..3 are read.  keep_out has capacity boxes_num and receives positions in the sorted array.
 *     Suppression is `IoU > thresh` (strict, nms_kernel.cu:71) with +1 widths; results are bit-exact with the
 *     reference.  boxes_num == 0 is legal (num_out = 0).  Synchronous.  max_keep < 0 means "all".
 * ------------------------------------------------------------------------------------------------------------- */
MNC_API int mnc_nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                    float nms_overlap_thresh, int device_id);
/* Same, but stops after max_keep survivors (ProposalLayer only uses keep[:300], proposal_layer.py:151-153);
 * the first max_keep indices are identical to the unbounded call. */
MNC_API int mnc_nms_topk(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                         float nms_overlap_thresh, int max_keep, int device_id);
/* Batched form for gpu_mask_voting's per-class loop (lib/transform/mask_transform.py:228-240): `batch` NMS problems
 * over ONE box set.  order_host: [batch][boxes_num] int32, item b's boxes in descending score order (indices into
 * boxes_host, i.e. the `argsort()[::-1]` gpu_nms.pyx:26 computes per call).  keep_out: [batch][boxes_num] int32 positions in
 * item b's order (first num_out[b] valid); num_out: [batch].  Each item's result is bit-identical to mnc_nms_topk on
 * boxes_host[order_host[b]].  One mask launch + one scan launch + one copy each way. */
MNC_API int mnc_nms_batched(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                            const int* order_host, int batch, float nms_overlap_thresh, int max_keep, int device_id);
/* The raw 64x64-tiled suppression bitmask (nms_kernel.cu:34-78) for word-for-word parity tests:
 * mask_host has boxes_num * ceil(boxes_num/64) uint64 words, row-major.  Lower-triangle words (never read by the
 * scan, nms_kernel.cu:135) are written as 0. */
MNC_API int mnc_nms_mask(unsigned long long* mask_host, const float* boxes_host, int boxes_num, int boxes_dim,
                         float nms_overlap_thresh, int device_id);
/* The reference's own symbol, twice.
 *  (1) C++ linkage -- `_Z4_nmsPiS_PKfiifi` -- which is what the reference's extension links: gpu_nms.pyx:13-14 declares it with
 *      `cdef extern from "gpu_nms.hpp"` and lib/setup.py:126-130 compiles that extension with language='c++', so the call is a
 *      C++ call of `void _nms(int*, int*, const float*, int, int, float, int)` (gpu_nms.hpp:1-2).  libmnc_hip.so exports that
 *      mangled name (csrc/ref_cxx_abi.hip); a C++ translation unit gets the declaration by including the reference's
 *      gpu_nms.hpp itself, or this header with MNC_HIP_REF_CXX_NAMES defined.
 *  (2) C linkage `_nms`, same arguments, for dlsym / ctypes / cgo callers (the default declaration of this header).
 * Both report errors through mnc_last_error() and a line on stderr, and set *num_out = 0 (the reference aborts in CUDA_CHECK). */
#if defined(__cplusplus) && defined(MNC_HIP_REF_CXX_NAMES)
}  /* leave extern "C" for the two C++-linkage names */
MNC_API void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                  float nms_overlap_thresh, int device_id);
MNC_API void _mv(const float* all_boxes, const float* all_masks, const int all_boxes_num, const int* candidate_inds,
                 const int* candidate_start, const float* candidate_weights, const int candidate_num,
                 const int image_height, const int image_width, const int box_dim, const int mask_size,
                 const int result_num, float* finalize_output_mask, int* finalize_output_box, const int device_id);
extern "C" {
#else
MNC_API void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                  float nms_overlap_thresh, int device_id);
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * b2  nms.mv  --  replaces `_mv` (lib/nms/gpu_mv.hpp:1-4, lib/nms/mv_kernel.cu:242-348), called from
 *     gpu_mv.pyx:13-31.  Same 15 arguments, same meaning.  candidate_start[r] is the END offset of result r
 *     (mask_transform.py:268).  finalize_output_mask: [result_num][mask_size][mask_size] float32;
 *     finalize_output_box: [result_num][4] int32 (x1,y1,x2,y2).  result_num == 0 or candidate_num == 0 is legal.
 *     The render -> aggregate -> reduce -> resize chain is fused; the N*H*W render buffer is never materialised.
 *     Results are bit-exact with the reference kernels evaluated without FMA contraction.  device_id IS honoured
 *     (the reference ignores it).
 * ------------------------------------------------------------------------------------------------------------- */
MNC_API int mnc_mv(const float* all_boxes, const float* all_masks, int all_boxes_num, const int* candidate_inds,
                   const int* candidate_start, const float* candidate_weights, int candidate_num, int image_height,
                   int image_width, int box_dim, int mask_size, int result_num, float* finalize_output_mask,
                   int* finalize_output_box, int device_id);
/* `_mv`: exported with C++ linkage (`_Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii`, what gpu_mv.pyx:7-8 + lib/setup.py:143-147 link)
 * and with C linkage, as `_nms` above. */
#if !(defined(__cplusplus) && defined(MNC_HIP_REF_CXX_NAMES))
MNC_API void _mv(const float* all_boxes, const float* all_masks, const int all_boxes_num, const int* candidate_inds,
                 const int* candidate_start, const float* candidate_weights, const int candidate_num,
                 const int image_height, const int image_width, const int box_dim, const int mask_size,
                 const int result_num, float* finalize_output_mask, int* finalize_output_box, const int device_id);
#endif

/* gpu_mask_voting in ONE call (lib/transform/mask_transform.py:213-286): per-class NMS (batched on the device) -> global
 * score threshold -> candidate sets {IoU_f64 >= iou_thresh} with class-score weights divided by float32(sequential float64 sum)
 * (python's sum() under the numpy 1.x the reference ran on, :266) -> fused mask voting kernels.  All host pointers.
 *   boxes [n][4] f32 (original-image pixels), masks [n][S][S] f32, scores [n][num_classes] f32 (column 0 = background),
 *   order [num_classes-1][n] i32: for class c+1, box indices by descending scores[:,c+1] (the caller's argsort()[::-1],
 *         gpu_nms.pyx:26), or NULL: the library orders each class itself (on the device) exactly as
 *         np.argsort(-scores[:, c+1], kind="stable") does -- ties in index order, NaN last.
 * Order, NMS, the global threshold, result rows, candidate sets (double-precision IoU) and voting all run on the device as one
 * asynchronous launch sequence; the host copies the inputs up and the result records down.
 * Limit: (num_classes-1) * min(max_per_image, n) <= 8192 kept boxes.
 * Outputs (capacity (num_classes-1)*min(max_per_image, n) rows): out_mask [R][S][S], out_box [R][4] i32, out_score [R],
 * class_count [num_classes-1] (rows per class, in class order), *result_num = R.
 * Bit-identical to running nms.gpu_nms x (num_classes-1), utils.cython_bbox.bbox_overlaps and nms.mv.mv as the reference does. */
MNC_API int mnc_mask_voting(const float* boxes, const float* masks, const float* scores, const int* order, int n,
                            int num_classes, int mask_size, int max_per_image, float nms_thresh, float iou_thresh,
                            int image_height, int image_width, float* out_mask, int* out_box, float* out_score,
                            int* class_count, int* result_num, int device_id);

/* ---------------------------------------------------------------------------------------------------------------
 * b3  utils.cython_bbox.bbox_overlaps (lib/utils/bbox.pyx:15-55): float64 IoU with +1 widths, [N][K] row-major.
 *     A host function in the reference (Cython) and here (C); it is not a GPU kernel and has no GPU counterpart.
 * ------------------------------------------------------------------------------------------------------------- */
MNC_API int mnc_bbox_overlaps(const double* boxes, int n, const double* query_boxes, int k, double* overlaps);

/* ---------------------------------------------------------------------------------------------------------------
 * Engine context: one per process per GPU (b5: `caffe.set_device`, one `caffe.Net` per process).  Owns a HIP
 * stream and all device scratch.  Not thread-safe; use one context per thread.
 * Hosts that keep several images in flight use one context (stream) per image.  The HIP runtime maps streams onto
 * GPU_MAX_HW_QUEUES hardware queues (default 4; streams that share a queue serialise): when this library is loaded it sets
 * GPU_MAX_HW_QUEUES=16 in the process environment unless the variable is already set -- effective when that happens before the
 * process's first HIP call (round 6; profiles/r06_streams.txt).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mnc_ctx mnc_ctx;

MNC_API int mnc_ctx_create(mnc_ctx** out, int device_id);
MNC_API int mnc_ctx_destroy(mnc_ctx* ctx);
MNC_API int mnc_ctx_sync(mnc_ctx* ctx);
MNC_API int mnc_ctx_device(const mnc_ctx* ctx, int* device_id);
/* Number of times one of the context's internal device arenas (split-K / Winograd scratch, proposal state, voting scratch) has
 * been re-allocated.  A captured HIP graph holds their addresses: mnc_forward_image drops its graph when this value has moved
 * since the capture; a caller that captures library launches into its own graph must do the same. */
MNC_API int mnc_ctx_arena_generation(const mnc_ctx* ctx, unsigned long* generation);

/* Conventions of the three Caffe layers whose source (caffe-mnc) is not available: ROIWarping, MaskResize, MaskPooling
 * (models/VGG16/mnc_5stage/test.prototxt:479-492, 558-567, 631-637, 809-820).  Every SPEC-CHOICE of oracle/SPEC.md is a field;
 * all fields zero (maskpool_thresh unused) IS the SPEC and the default of every context -- PARITY UNPINNED either way.  A user who
 * can read caffe-mnc (or holds the published mnc_model.caffemodel.h5 and an image with known output) selects the matching
 * convention here / in mnc_net_config / with `caffe.Net(..., layer_conventions={...})` / cfg.LAYER_CONVENTIONS; nothing is
 * recompiled.  The CPU oracle evaluates the same switches (oracle/mnc_oracle.c: orc_roi_warp_ex, orc_mask_resize_ex,
 * orc_mask_pool_ex), bit for bit; oracle/SPEC.md section 6 tabulates how far each alternative moves the outputs.
 * Read by mnc_roi_warp[_sm], mnc_mask_resize, mnc_mask_pool[_sm], mnc_box_mask_pool at launch time. */
typedef struct mnc_layer_conventions {
  int warp_sample;       /* ROIWarping sample position in bin g: 0 x1s + g*bin (top-left, SPEC) | 1 x1s + (g+0.5)*bin (bin centre)
                          * | 2 x1s + (g+0.5)*bin - 0.5 (bin centre, pixel-centre coordinates) */
  int warp_round_edges;  /* 0 scaled RoI edges un-rounded (SPEC) | 1 floor(x*scale + 0.5), as ROIPooling */
  int warp_no_plus_one;  /* 0 roi_w = max(x2s - x1s + 1, 1) (SPEC) | 1 roi_w = max(x2s - x1s, 1) */
  int warp_oob;          /* 0 bilinear taps outside the map contribute 0 (SPEC) | 1 taps clamped to the border */
  int resize_mode;       /* MaskResize source position: 0 dst*in/out, nearest on the last row/column (mv_kernel.cu:193-240, SPEC)
                          * | 1 (dst+0.5)*in/out - 0.5 (half-pixel centres) | 2 dst*(in-1)/(out-1) (align_corners) */
  int maskpool_binary;   /* MaskPooling: 0 feature * continuous mask (SPEC) | 1 feature * (mask >= maskpool_thresh) */
  float maskpool_thresh; /* 0.4 = cfg.BINARIZE_THRESH */
  int inherit;           /* read by mnc_net_create only (mnc_net_config.conventions): 1 = leave the conventions in force on the
                          * context alone (what mnc_net_default_config sets), 0 = apply this struct to the context -- all other
                          * fields zero then selects the SPEC explicitly, whatever an earlier net or the host had set.
                          * mnc_ctx_set_layer_conventions ignores it. */
} mnc_layer_conventions;
MNC_API int mnc_ctx_set_layer_conventions(mnc_ctx* ctx, const mnc_layer_conventions* conv);   /* NULL: back to the SPEC */
/* Override one of the launchers' own choices on this context (a tile shape, a kernel variant, a plan switch): name as in
 * csrc/mnc_internal.h MNC_TUNE_KEYS, e.g. "FC_TILE" / "5", "CONV1X1_TILE" / "2,4"; value NULL or "" = the library's choice again.
 * The environment variable MNC_<NAME> sets the same value when the context is created -- launch paths never read the
 * environment.  For tests (every variant reachable at a small shape) and A/B measurements; results never change beyond the
 * summation grouping of K splits. */
MNC_API int mnc_ctx_set_tuning(mnc_ctx* ctx, const char* name, const char* value);
MNC_API int mnc_ctx_get_layer_conventions(const mnc_ctx* ctx, mnc_layer_conventions* conv);

/* Launch-sequence capture for hosts that drive the per-layer entry points themselves (the caffe-shaped Python engine does, for
 * any prototxt): everything the library enqueues on the context's stream between capture_begin and capture_end -- kernels,
 * mnc_h2d_async / mnc_d2h_async / mnc_d2d copies -- becomes one HIP graph; mnc_graph_launch replays it with one call.  Inside a
 * capture nothing may synchronise or allocate (mnc_h2d, mnc_d2h, mnc_ctx_sync, mnc_dev_alloc / _free, a growing internal
 * arena): such a call is refused with MNC_ERR_STATE BEFORE it touches the stream (the capture stays intact: end it, discard the
 * graph, run the sequence eagerly once so that every buffer has its size, capture again).  mnc_ctx_capture_begin itself returns
 * MNC_ERR_STATE while per-launch profiling is on (mnc_prof_enable): event pairs cannot be captured.  The graph holds raw device addresses: mnc_graph_launch returns MNC_ERR_STATE when an internal arena of
 * the context has been re-allocated since the capture (mnc_ctx_arena_generation) -- capture again.  The caller keeps its own
 * buffers in place.  mnc_forward_image uses the same mechanism internally. */
typedef struct mnc_graph mnc_graph;
MNC_API int mnc_ctx_capture_begin(mnc_ctx* ctx);
MNC_API int mnc_ctx_capture_end(mnc_ctx* ctx, mnc_graph** out);     /* *out = NULL and an error status when the capture failed */
MNC_API int mnc_graph_launch(mnc_ctx* ctx, mnc_graph* graph);        /* asynchronous on the context's stream */
MNC_API int mnc_graph_destroy(mnc_graph* graph);
/* Device address of the row count the last mnc_proposal left on the device (valid until the proposal state is re-allocated):
 * lets a captured sequence copy it down with its results instead of calling mnc_proposal_count (which synchronises). */
MNC_API int mnc_proposal_count_ptr(mnc_ctx* ctx, void** d_count);

/* Device memory for the host-side executor (the caffe-shaped Net keeps its blobs here). */
MNC_API int mnc_dev_alloc(mnc_ctx* ctx, size_t bytes, void** d_ptr);
MNC_API int mnc_dev_free(mnc_ctx* ctx, void* d_ptr);
MNC_API int mnc_h2d(mnc_ctx* ctx, void* d_dst, const void* src_host, size_t bytes);   /* stream-ordered + sync */
MNC_API int mnc_d2h(mnc_ctx* ctx, void* dst_host, const void* d_src, size_t bytes);   /* stream-ordered + sync */
MNC_API int mnc_d2d(mnc_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);      /* stream-ordered, async */
MNC_API int mnc_dev_zero(mnc_ctx* ctx, void* d_ptr, size_t bytes);
/* Page-locked host memory for the image that goes up and the instance records that come down, and the stream-ordered copies
 * that do NOT synchronise (from / to pinned memory they are truly asynchronous; the caller synchronises with mnc_ctx_sync). */
MNC_API int mnc_host_alloc(mnc_ctx* ctx, size_t bytes, void** host_ptr);
MNC_API int mnc_host_free(mnc_ctx* ctx, void* host_ptr);
MNC_API int mnc_h2d_async(mnc_ctx* ctx, void* d_dst, const void* src_host, size_t bytes);
MNC_API int mnc_d2h_async(mnc_ctx* ctx, void* dst_host, const void* d_src, size_t bytes);

/* Per-kernel timing with HIP events on the context's stream (bench.py's `roofline` numbers come from here).
 * enable=1 records a start/stop event pair around every kernel launched through the context; enable=2 only around the
 * launches that carry >= 1 GFLOP of algorithmic work (the MFMA kernels), which keeps the timed region almost undisturbed. */
MNC_API int mnc_prof_enable(mnc_ctx* ctx, int enable);
MNC_API int mnc_prof_reset(mnc_ctx* ctx);
MNC_API int mnc_prof_count(mnc_ctx* ctx, int* n_records);                 /* synchronises the stream */
MNC_API int mnc_prof_get(mnc_ctx* ctx, int index, char* name_buf, int name_cap, float* ms, double* flops,
                         double* bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Layout conversion and weight packing (device -> device, asynchronous on the context's stream).
 * ------------------------------------------------------------------------------------------------------------- */
MNC_API int mnc_nchw_to_c8(mnc_ctx* ctx, const float* d_nchw, float* d_c8, int C, int H, int W);  /* C%8==0 */
MNC_API int mnc_c8_to_nchw(mnc_ctx* ctx, const float* d_c8, float* d_nchw, int C, int H, int W);
/* [R][C][PH][PW] (Caffe) <-> [R][PH][PW][C] (engine) */
MNC_API int mnc_rchw_to_rhwc(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int C, int PH, int PW);
MNC_API int mnc_rhwc_to_rchw(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int C, int PH, int PW);
/* Caffe conv weight [Cout][Cin][3][3] -> packed [Cin/8][Cout][76].  Cin%8==0, Cout%32==0.  Elements: Cin/8*Cout*76 */
MNC_API int mnc_pack_conv3x3_weights(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin);
/* Caffe InnerProduct weight [N][C*PH*PW] (columns in (c,h,w) order) -> [N][PH*PW*C] ((h,w,c) order). */
MNC_API int mnc_pack_fc_weights(mnc_ctx* ctx, const float* d_nchw_cols, float* d_hwc_cols, int N, int C, int PH, int PW);

/* ---------------------------------------------------------------------------------------------------------------
 * Graph ops -- one per layer type of models/VGG16/mnc_5stage/test.prototxt.  All asynchronous on the
 * context's stream; all tensors fp32.
 * ------------------------------------------------------------------------------------------------------------- */
/* conv1_1 (test.prototxt:19-40): 3x3 pad 1, Cin = 3, reads the NCHW input blob, + bias + ReLU -> c8.  HBM-bound. */
MNC_API int mnc_conv3x3_c3(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias,
                           float* d_out_c8, int H, int W, int Cout, int relu);
/* Convolution 3x3 pad 1 stride 1 + bias (+ ReLU) (test.prototxt:41-412), c8 -> c8, fp32 MFMA implicit GEMM. */
MNC_API int mnc_conv3x3(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias,
                        float* d_out_c8, int H, int W, int Cin, int Cout, int relu);
/* The same convolution on the bf16 matrix pipe with fp32-class accuracy ("bf16x3", see mnc_fc_bf16x3 and
 * mnc_amd/csrc/conv_sw.hip; BASELINE.json configs[2] "bf16 convs via MFMA").  Activations fp32 c8 in and out at this entry point
 * (packed forms: "Reduced-precision 3x3 convolutions" below); d_w_packed comes from mnc_pack_conv3x3_bf16x3 =
 * mnc_pack_conv3x3_lowp(mode 0), mnc_conv3x3_lowp_weight_bytes(0, Cout, Cin) bytes.  Cin%8==0, Cout%32==0. */
MNC_API int mnc_pack_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin);
MNC_API int mnc_conv3x3_bf16x3(mnc_ctx* ctx, const float* d_in_c8, const void* d_w_packed, const float* d_bias,
                               float* d_out_c8, int H, int W, int Cin, int Cout, int relu);
/* The same convolution (fp32 in, fp32 out, fp32 MFMA) by Winograd's minimal filtering F(2x2, 3x3): 16 instead of 36 multiplies
 * per (input channel, output channel, 2x2 output tile) -- 2.25x fewer matrix-pipe cycles (mnc_amd/csrc/conv_wino.hip).  All
 * transform coefficients are 0, +-1, +-1/2: input / output transforms are exact fp32 additions, the filter transform is evaluated
 * in double and rounded once; results agree with mnc_conv3x3 to fp32 rounding (not bit for bit: different summation order).
 * d_w_packed from mnc_pack_conv3x3_wino: Caffe [Cout][Cin][3][3] -> [Cin/8][Cout/32][2][32][68] floats (Cin*Cout*17 floats). */
MNC_API int mnc_pack_conv3x3_wino(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin);
MNC_API int mnc_conv3x3_wino(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias, float* d_out_c8,
                             int H, int W, int Cin, int Cout, int relu);
/* The same followed by the Pooling MAX 2x2 stride 2 of the trunk (test.prototxt:69-79, 130-140, 216-226, 302-312) in the kernel's
 * epilogue: a 2x2 Winograd output tile IS a pooling window, so the pooled value is the maximum of a lane's own outputs and the
 * full-resolution tensor never reaches HBM.  d_out_pooled_c8: [Cout/8][ceil(H/2)][ceil(W/2)][8] (Caffe's ceil output size). */
MNC_API int mnc_conv3x3_wino_pool(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias,
                                  float* d_out_pooled_c8, int H, int W, int Cin, int Cout, int relu);
/* The same convolution by Winograd's F(4x4, 3x3) (round 4; mnc_amd/csrc/conv_wino4.hip): 36 multiplies per (input channel, output
 * channel, 4x4 output tile) = 2.25 per output against F(2x2)'s 4 and the direct form's 9.  Fused: input transform, the 36 channel
 * contractions (v_mfma_f32_16x16x4_f32) and the output transform run in one kernel; four-wave workgroups (32 channels x 8 x 64
 * pixels) with half of a CU's LDS each, two per CU; the two waves of a tile row split the 36 positions (half of the packed-fp32
 * input transform each) and exchange partial output sums through LDS in the epilogue.  The transforms carry the coefficients 2, 4, 5, 8 and 1/6, 1/12, 1/24 (filter side, evaluated in double,
 * rounded once): rounding error ~1e-5 of the output range at 512 input channels (F(2x2): ~1e-6; both inside the kernels' 1e-4
 * bar).  d_w_packed from mnc_pack_conv3x3_wino4: Caffe [Cout][Cin][3][3] -> [Cin/8][Cout/32][2][2][64][36] floats (Cin*Cout*36 floats).
 * Cin%8==0, Cout%32==0; the input tensor and the packed weights each below 2 GB (32-bit buffer offsets; beyond that mnc_conv3x3_wino).  _pool: the following Pooling MAX 2x2/2 in the epilogue (a 4x4 tile is four windows). */
MNC_API int mnc_pack_conv3x3_wino4(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin);
MNC_API int mnc_conv3x3_wino4(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias, float* d_out_c8,
                              int H, int W, int Cin, int Cout, int relu);
MNC_API int mnc_conv3x3_wino4_pool(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias,
                                   float* d_out_pooled_c8, int H, int W, int Cin, int Cout, int relu);
/* Pooling MAX 2x2 stride 2 with Caffe's ceil output size (test.prototxt:69-79,...): c8 [C/8][H][W][8] ->
 * [C/8][OH][OW][8], OH = ceil((H-2)/2)+1. */
MNC_API int mnc_maxpool2_c8(mnc_ctx* ctx, const float* d_in, float* d_out, int C, int H, int W);
/* rpn_cls_score / rpn_bbox_pred (test.prototxt:413-439): 1x1 conv c8 -> NCHW [Cout][H][W], weight [Cout][Cin]. */
MNC_API int mnc_conv1x1_to_nchw(mnc_ctx* ctx, const float* d_in_c8, const float* d_w, const float* d_bias,
                                float* d_out_nchw, int H, int W, int Cin, int Cout);
/* Reshape(0,2,-1,0) -> Softmax(axis 1) -> Reshape(0,18,-1,0) (test.prototxt:440-462): pairs channel a with A+a. */
MNC_API int mnc_rpn_softmax(mnc_ctx* ctx, const float* d_score_nchw, float* d_prob_nchw, int A, int H, int W);
/* rpn_cls_score + rpn_bbox_pred + the softmax above in ONE launch (test.prototxt:413-462): d_w = [2A cls rows | 4A bbox rows] x
 * [Cin] (the two layers' weights concatenated, biases likewise), d_score = the 6A score planes (NCHW: the first 2A are
 * rpn_cls_score, the last 4A rpn_bbox_pred), d_prob = the 2A probability planes.  The same bits as mnc_conv1x1_to_nchw on the
 * concatenated weights followed by mnc_rpn_softmax(A). */
MNC_API int mnc_rpn_heads(mnc_ctx* ctx, const float* d_in_c8, const float* d_w, const float* d_bias, float* d_score_nchw,
                          float* d_prob_nchw, int H, int W, int Cin, int A);
/* ROIWarping (test.prototxt:479-492, 809-820) per oracle/SPEC.md section 1, c8 feature -> [R][PH][PW][C].
 * pool2 != 0 fuses the following Pooling MAX 2x2/2 (test.prototxt:494-505): the warp is evaluated at
 * 2PH x 2PW and max-reduced, so the 28x28 "premax" tensor never reaches HBM. */
MNC_API int mnc_roi_warp(mnc_ctx* ctx, const float* d_feat_c8, int C, int H, int W, const float* d_rois, int R,
                         int PH, int PW, float spatial_scale, int pool2, float* d_out_rhwc);
/* ---- general convolution / pooling / residual ops (graphs beyond VGG-16, e.g. a ResNet-50 trunk;
 * BASELINE.json configs[4]).  Public BVLC Caffe semantics (convolution_param / pooling_param / eltwise_param); there is no
 * such model in the reference repository. ---- */
/* Caffe conv weight [Cout][Cin][KH][KW] -> [KH*KW][Cin/8][Cout][8] for mnc_conv2d.  Cin%8==0, Cout%8==0; same element count. */
MNC_API int mnc_pack_conv_weights(mnc_ctx* ctx, const float* d_oihw, float* d_packed, int Cout, int Cin, int KH, int KW);
/* Convolution, any kernel / stride / pad, c8 -> c8 ([Cout/8][OH][OW][8], OH = (H + 2 pad - KH)/stride + 1), fp32 MFMA implicit
 * GEMM: out = conv(in) + bias (+ d_residual, same layout as out, may be NULL) (+ ReLU).  A BatchNorm + Scale pair behind the
 * convolution is folded into the weights and bias by the caller. */
MNC_API int mnc_conv2d(mnc_ctx* ctx, const float* d_in_c8, const float* d_w_packed, const float* d_bias,
                       const float* d_residual_c8, float* d_out_c8, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                       int pad, int relu);
/* "f16" math mode of mnc_conv2d: activations rounded to fp16 while staged, weights from mnc_pack_conv_weights_f16
 * ([KH*KW][ceil(Cin/32)][Cout][32] halves, channel groups zero-padded: ceil(Cin/32)*32*Cout*KH*KW*2 bytes), fp32 accumulate,
 * same epilogue. */
MNC_API int mnc_pack_conv_weights_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin, int KH, int KW);
MNC_API int mnc_conv2d_f16(mnc_ctx* ctx, const float* d_in_c8, const void* d_w_packed, const float* d_bias,
                           const float* d_residual_c8, float* d_out_c8, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int relu);
/* 1x1 convolution, stride 1 or 2, no padding, as a plain GEMM (csrc/conv1x1.hip): the c8 layout is already the MFMA B
 * fragment order, so both operands go global -> registers -> matrix pipe with no LDS staging.  Same epilogue as mnc_conv2d
 * (+ bias, + residual, ReLU).  Weights from mnc_pack_conv1x1: Caffe [Cout][Cin] fp32 -> A-fragment order
 * [K-steps][ceil(Cout/32)][64 lanes] x 16 bytes (f16 != 0: halves, K-step 16, Cin%16==0; else fp32, K-step 8):
 * Cin * ceil(Cout/32)*32 * (f16 ? 2 : 4) bytes.
 * mnc_conv1x1: fp32 c8 in / residual / out on the fp32 matrix pipe.
 * mnc_conv1x1_f16_pk ("f16" math mode): d_in is the packed fp16 c8 tensor ([C/8][H][W][8] halves, see "2-byte activation
 * tensors" below); d_out packed fp16 (out_packed != 0; nearest-even rounding of the fp32 result) or fp32 c8; d_residual
 * (may be NULL) packed fp16 (res_packed != 0) or fp32 c8; fp32 accumulate. */
MNC_API int mnc_pack_conv1x1(mnc_ctx* ctx, const float* d_w, void* d_packed, int Cout, int Cin, int f16);
MNC_API int mnc_conv1x1(mnc_ctx* ctx, const float* d_in_c8, const void* d_w_packed, const float* d_bias,
                        const float* d_residual_c8, float* d_out_c8, int H, int W, int Cin, int Cout, int stride, int relu);
MNC_API int mnc_conv1x1_f16_pk(mnc_ctx* ctx, const void* d_in_pk, const void* d_w_packed, const float* d_bias,
                               const void* d_residual, void* d_out, int H, int W, int Cin, int Cout, int stride, int relu,
                               int res_packed, int out_packed);
/* First convolution of a 3-channel NCHW input blob (ResNet conv1 7x7/2 pad 3): weights [Cout][3][K][K] as in Caffe,
 * + bias (+ ReLU) -> c8.  Cout%16==0.  mnc_conv_stem_c3_fmt: out_packed != 0 writes the packed fp16 c8 tensor instead. */
MNC_API int mnc_conv_stem_c3(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias, float* d_out_c8,
                             int H, int W, int Cout, int K, int stride, int pad, int relu);
MNC_API int mnc_conv_stem_c3_fmt(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias, void* d_out,
                                 int H, int W, int Cout, int K, int stride, int pad, int relu, int out_packed);
/* "f16" math mode of the stem: the same convolution on the fp16 matrix pipe (csrc/conv_gen.hip: the kernel rows padded to 8 taps,
 * so a lane's 8 K-values are 8 consecutive input pixels read straight from the NCHW blob; weights in registers).  Input rounded
 * to fp16 in registers, weights once by mnc_pack_conv_stem_f16 ([ceil(3K/2)][Cout/32][64] x 16 bytes), fp32 accumulate; output
 * fp32 c8 or packed fp16 c8.  K = 3, 5 or 7; Cout%32==0. */
MNC_API int mnc_pack_conv_stem_f16(mnc_ctx* ctx, const float* d_w_oihw, void* d_packed, int Cout, int K);
MNC_API int mnc_conv_stem_f16(mnc_ctx* ctx, const float* d_in_nchw, const void* d_w_packed, const float* d_bias, void* d_out, int H,
                              int W, int Cout, int K, int stride, int pad, int relu, int out_packed);
/* Pooling MAX with any kernel / stride / pad on a c8 map; Caffe's ceil output size, windows clipped to the image.
 * mnc_maxpool_c8_f16: the same on the packed fp16 c8 tensor (max commutes with the rounding: bit for bit the fp16 form of
 * the fp32 result). */
MNC_API int mnc_maxpool_c8(mnc_ctx* ctx, const float* d_in, float* d_out, int C, int H, int W, int K, int stride, int pad);
MNC_API int mnc_maxpool_c8_f16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W, int K, int stride, int pad);
/* Eltwise SUM of two tensors of the same layout (+ ReLU): d_out[i] = d_a[i] + d_b[i]. */
MNC_API int mnc_add(mnc_ctx* ctx, const float* d_a, const float* d_b, float* d_out, size_t n, int relu);
/* prep_im_for_blob (lib/utils/blob.py:36-50) and one level of prep_im_for_blob_cfm (:53-85) on the device: a uint8 BGR image
 * [H][W][3] -> float32 planes [3][PH][PW] holding (pixel - mean) resized with cv2.resize's INTER_LINEAR rule to OH x OW and
 * zero-padded to the blob's PH x PW (im_list_to_blob, :17-33).  `means` = 3 host doubles (cfg.PIXEL_MEANS).  The resize
 * taps come from the caller (the host function that the numpy path uses): d_x0[OW] / d_y0[OH] first source index,
 * d_ax[OW] / d_ay[OH] fraction of the next one; the second index is min(first + 1, size - 1).  Bit-identical to the host
 * path (tests/test_gpu_ops.py). */
MNC_API int mnc_prep_image(mnc_ctx* ctx, const unsigned char* d_bgr_hwc, int H, int W, const double* means_host,
                           const int* d_x0, const float* d_ax, int OW, const int* d_y0, const float* d_ay, int OH,
                           float* d_out_chw, int PH, int PW);
/* ROIPooling (models/VGG16/cfm/test.prototxt:397-407 7x7, :446-456 14x14; the Fast R-CNN layer of the absent caffe-mnc
 * submodule, restated in oracle/SPEC.md section 4): max over the integer bins of round(roi * spatial_scale).  The feature
 * is a batch of N c8 images [N][C/8][H][W][8] (CFM feeds an image pyramid, lib/caffeWrapper/TesterWrapper.py:371-399);
 * rois are [R][5] = (batch index, x1, y1, x2, y2); output [R][PH][PW][C].  Empty bins give 0. */
MNC_API int mnc_roi_pool(mnc_ctx* ctx, const float* d_feat_c8, int N, int C, int H, int W, const float* d_rois, int R,
                         int PH, int PW, float spatial_scale, float* d_out_rhwc);
/* Pooling MAX 2x2/2 on per-RoI features [R][PH][PW][C] -> [R][PH/2][PW/2][C] (test.prototxt:571-582,...). */
MNC_API int mnc_maxpool2_rhwc(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int PH, int PW, int C);
/* MaskResize (test.prototxt:558-567) per SPEC.md section 2: [R][IH][IW] -> [R][OH][OW]. */
MNC_API int mnc_mask_resize(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int IH, int IW, int OH, int OW);
/* MaskPooling (test.prototxt:631-637) per SPEC.md section 3: feat[R][PH][PW][C] * mask[R][PH][PW].
 * pool2 != 0 fuses the following Pooling MAX 2x2/2 (test.prototxt:639-650). */
MNC_API int mnc_mask_pool(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_out, int R, int PH,
                          int PW, int C, int pool2);
/* InnerProduct (+ReLU / +Sigmoid): out[M][N] = act(A[M][K] . W[N][K]^T + bias[N]).  fp32 MFMA, split-K chosen
 * internally; d_out may be a column slice of a wider matrix (ldc >= N) so Concat (test.prototxt:700-709) is free.
 * act: 0 none, 1 ReLU, 2 sigmoid. */
MNC_API int mnc_fc(mnc_ctx* ctx, const float* d_a, const float* d_w, const float* d_bias, float* d_out, int M,
                   int N, int K, int ldc, int act);
/* Two InnerProducts of ONE shape in one launch: out_i = act(a_i . w_i^T + bias_i), i = 0, 1 (the box and the mask branch of a head
 * stage: fc6 + fc6_mask, fc7 + fc7_mask, test.prototxt:584-627 / :652-696).  Twice the column tiles fill the chip with half as
 * many K ranges: longer ranges per workgroup, half the partial sums.  Shapes mnc_fc would not give to its 320-row kernel in one
 * launch run as two mnc_fc calls.  The paired launch groups the partial sums differently from mnc_fc (results differ in the last
 * bits): every executor of a graph pairs the same layers. */
MNC_API int mnc_fc_pair(mnc_ctx* ctx, const float* d_a0, const float* d_w0, const float* d_bias0, float* d_out0, const float* d_a1,
                        const float* d_w1, const float* d_bias1, float* d_out1, int M, int N, int K, int ldc, int act);
/* InnerProduct on the bf16 matrix pipe with fp32-class accuracy ("bf16x3": every operand split into hi + lo bf16, product =
 * a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate; relative error ~1e-5 per product, see mnc_amd/csrc/gemm_x3.hip).
 * d_w_packed comes from mnc_pack_fc_bf16x3: fp32 [N][K] -> stage-major tiles [ceil(N/128)][K/32][128][(hi x8 | lo x8) x 4] bf16,
 * ceil(N/128)*128*K*4 bytes (rows past N are zero), K%32==0 -- the weight panel a workgroup needs for one K stage is one
 * contiguous 16 KB.  Activations are fp32 at the interface (split per call into the context's scratch arena).
 * Same contract as mnc_fc otherwise. */
MNC_API int mnc_pack_fc_bf16x3(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K);
MNC_API int mnc_fc_bf16x3(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M,
                          int N, int K, int ldc, int act);
/* Reduced-precision 3x3 convolutions (round 6: csrc/conv_sw.hip -- sliding-window implicit GEMM on 2-byte activation planes, every
 * operand by LDS-DMA, K ranges summed inside the workgroup; models/VGG16/mnc_5stage/test.prototxt:41-412).  mode: 0 = bf16x3 (split
 * precision, above), 1 = f16 (one fp16 product per term on v_mfma_f32_32x32x16_f16, operands rounded to nearest even, fp32
 * accumulation), 2 = bf16 (the same with bf16: BASELINE configs[2] "bf16 convs via MFMA" as written; ~4e-3 of a layer's range per
 * layer, measured and recorded, outside the 1e-3 bar -- bf16x3 is the bf16-pipe mode that keeps it).
 * Packed weights: mnc_pack_conv3x3_lowp (and the per-mode names of rounds 1-5, which call it) writes
 *   [ceil(Cin/16)][Cout/32][planes][9 taps][32 channels] x 16 B (8 two-byte values)
 * planes = the two 8-channel halves of the 16-channel block (f16 / bf16), or hi of each half then lo of each half (bf16x3: hi =
 * rne(w), lo = rne(w - hi)); channels past Cin are zero.  mnc_conv3x3_lowp_weight_bytes(mode, Cout, Cin) is the buffer's size:
 * ceil(Cin/16) * (Cout/32) * (mode == 0 ? 4 : 2) * 4608 bytes.
 * Packed 2-byte activations between MFMA layers.  A c8 tensor [C/8][H][W][8] is kept as
 *   bf16x3:  [C/8][H][W][hi x8 | lo x8] bf16 -- the split the kernels apply to an fp32 value (hi = truncation, lo = the
 *            remainder rounded half-up), two 2-byte planes interleaved per pixel, 32 B per pixel and channel block;
 *   f16:     [C/8][H][W][8] fp16 (round to nearest even), 16 B per pixel and channel block;
 *   bf16:    [C/8][H][W][8] bf16 (round to nearest even), 16 B (round 6).
 * The convolution multiplies PACKED inputs; an fp32 c8 input (in_packed = 0, and the fp32-tensor entry points) is packed into the
 * context's scratch arena first.  A producer's epilogue applies exactly that packing to its fp32 result, so a packed chain gives bit
 * for bit the results of the fp32-tensor chain (test.prototxt:41-412 is such a chain: conv1_1 .. conv5_3 with four MAX 2x2/2 pools).
 * mnc_conv3x3_lowp writes d_out_packed and / or d_out_c8 (fp32 c8); a null one is not written (conv5_3 feeds the RPN convolution
 * packed and the RoI warps in fp32: test.prototxt:395-424, 479-492).  The *_pk forms select one format per side.
 * mnc_maxpool2_c8_{bf16x3,f16,bf16}: Pooling MAX 2x2/2 (ceil output size) on the packed form; mnc_act_pack / mnc_act_unpack:
 * fp32 c8 <-> packed, n = element count (multiple of 8), f16 = the mode number (0 bf16x3, 1 f16, 2 bf16). */
MNC_API size_t mnc_conv3x3_lowp_weight_bytes(int mode, int Cout, int Cin);
MNC_API int mnc_pack_conv3x3_lowp(mnc_ctx* ctx, int mode, const float* d_oihw, void* d_packed, int Cout, int Cin);
MNC_API int mnc_conv3x3_lowp(mnc_ctx* ctx, int mode, const void* d_in_packed, const void* d_w_packed, const float* d_bias,
                             void* d_out_packed, float* d_out_c8, int H, int W, int Cin, int Cout, int relu);
/* The same convolution with the following Pooling MAX 2x2/2 (Caffe's ceil output size) folded into the epilogue: writes only the
 * pooled tensor [Cout/8][ceil(H/2)][ceil(W/2)] in the packed form -- bit for bit mnc_maxpool2_c8_* of the unpooled packed output
 * (conv1_2 / conv2_2 / conv3_3 / conv4_3 + pool1..4: test.prototxt:61-92, 117-148, 193-232, 277-316).  Cout % 64 == 0. */
MNC_API int mnc_conv3x3_lowp_pool(mnc_ctx* ctx, int mode, const void* d_in_packed, const void* d_w_packed, const float* d_bias,
                                  void* d_out_pooled_packed, int H, int W, int Cin, int Cout, int relu);
MNC_API int mnc_pack_conv3x3_f16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin);
MNC_API int mnc_pack_conv3x3_bf16(mnc_ctx* ctx, const float* d_oihw, void* d_packed, int Cout, int Cin);
MNC_API int mnc_conv3x3_bf16(mnc_ctx* ctx, const float* d_in_c8, const void* d_w_packed, const float* d_bias, float* d_out_c8,
                             int H, int W, int Cin, int Cout, int relu);
MNC_API int mnc_conv3x3_f16(mnc_ctx* ctx, const float* d_in_c8, const void* d_w_packed, const float* d_bias, float* d_out_c8,
                            int H, int W, int Cin, int Cout, int relu);
MNC_API int mnc_conv3x3_bf16x3_pk(mnc_ctx* ctx, const void* d_in, const void* d_w_packed, const float* d_bias, void* d_out,
                                  int H, int W, int Cin, int Cout, int relu, int in_packed, int out_packed);
MNC_API int mnc_conv3x3_f16_pk(mnc_ctx* ctx, const void* d_in, const void* d_w_packed, const float* d_bias, void* d_out,
                               int H, int W, int Cin, int Cout, int relu, int in_packed, int out_packed);
MNC_API int mnc_conv3x3_bf16_pk(mnc_ctx* ctx, const void* d_in, const void* d_w_packed, const float* d_bias, void* d_out,
                                int H, int W, int Cin, int Cout, int relu, int in_packed, int out_packed);
/* conv1_1 (mnc_conv3x3_c3) writing the packed form: out_fmt 0 = fp32 c8, 1 = bf16x3 packed, 2 = fp16 packed, 3 = bf16 packed */
MNC_API int mnc_conv3x3_c3_fmt(mnc_ctx* ctx, const float* d_in_nchw, const float* d_w_oihw, const float* d_bias, void* d_out,
                               int H, int W, int Cout, int relu, int out_fmt);
MNC_API int mnc_maxpool2_c8_bf16x3(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W);
MNC_API int mnc_maxpool2_c8_f16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W);
MNC_API int mnc_maxpool2_c8_bf16(mnc_ctx* ctx, const void* d_in, void* d_out, int C, int H, int W);
MNC_API int mnc_act_pack(mnc_ctx* ctx, const float* d_c8, void* d_packed, size_t n, int f16);
MNC_API int mnc_act_unpack(mnc_ctx* ctx, const void* d_packed, float* d_c8, size_t n, int f16);
/* "f16" math mode (BASELINE.json configs[4] names fp16): InnerProduct with both operands rounded to IEEE fp16 (nearest even)
 * and fp32 accumulation on v_mfma_f32_32x32x16_f16 -- one product per term, 2 bytes per value streamed instead of 4.
 * mnc_pack_fc_f16: Caffe weight [N][K] -> [ceil(N/128)][K/64][128][64] halves (bytes: ceil(N/128)*128*K*2), once at load.
 * mnc_fc_f16: same interface as mnc_fc (fp32 activations in, fp32 out); K%64==0.  Relative error vs fp32 ~3e-4 per layer. */
MNC_API int mnc_pack_fc_f16(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K);
/* The same InnerProduct in plain bf16 (nearest even, one product per term; the "bf16" math mode): mnc_fc_f16's layout and interface. */
MNC_API int mnc_pack_fc_bf16(mnc_ctx* ctx, const float* d_w, void* d_packed, int N, int K);
MNC_API int mnc_fc_bf16(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N,
                        int K, int ldc, int act);
MNC_API int mnc_fc_f16(mnc_ctx* ctx, const float* d_a, const void* d_w_packed, const float* d_bias, float* d_out, int M, int N,
                       int K, int ldc, int act);
/* Two reduced-precision InnerProducts of one shape as ONE launch (round 6; mode 1 = fp16, 2 = plain bf16): the box and the mask
 * branch of a head stage (fc6 + fc6_mask, fc7 + fc7_mask; test.prototxt:584-627, 652-696) -- mnc_fc_pair's idea on the 256-column
 * LDS-DMA kernel: twice the column tiles fill the chip with half the K ranges (half the partial sums: 2 x 39 MB instead of 2 x 79 at
 * fc6 / 300 RoIs).  Per product exactly the arguments of mnc_fc_f16_ex (one of d_a / d_a_sm; m_stride rows per stage of the
 * stage-major inputs; an optional second output in the next InnerProduct's form).  mode 0 (split bf16) and shapes the paired kernel
 * does not take (it needs 160 < M <= 320, N % 256 == 0, >= 8 stages per K range) run as the two single calls.  The paired launch
 * groups the partial sums differently from two single calls: every executor of a graph pairs the same layers. */
MNC_API int mnc_fc_lowp_pair(mnc_ctx* ctx, int mode, const float* d_a0, const void* d_a_sm0, const float* d_a1, const void* d_a_sm1,
                             int m_stride, const void* d_w0, const void* d_w1, const float* d_bias0, const float* d_bias1,
                             float* d_out0, float* d_out1, int M, int N, int K, int ldc, int act, void* d_out_sm0, void* d_out_sm1,
                             int out_sm_fmt);
/* ---- InnerProduct activations already in the reduced-precision kernels' own form ----
 * mnc_fc_bf16x3 / mnc_fc_f16 multiply the activations from a stage-major 2-byte tensor, which they otherwise make from the fp32
 * rows on every call (an elementwise pass over M x K: 0.2 ms per image at 300 RoIs, 0.75 ms at 1000 RoIs x 1024 channels):
 *   fmt 1 (f16)    [K/64][M][64] halves (nearest even of the fp32 value),                 M*K*2 bytes
 *   fmt 2 (bf16x3) [K/32][M][4 x (hi x8 | lo x8)] bf16 (x = hi + lo, both nearest even), M*K*4 bytes
 * with row r of the fp32 tensor [M][K] at row r of every stage.  The producers of the per-RoI tensors write this form next to the
 * fp32 tensor in their epilogue (the *_sm entry points below: d_sm may be NULL / sm_fmt 0 = no second output; values are bit for
 * bit what mnc_fc_pack_act makes of the fp32 output), and mnc_fc_{bf16x3,f16}_pre take it: m_stride = rows of the stage-major
 * tensor (>= M: a call may multiply its first M rows), everything else as mnc_fc_*.
 *   mnc_roi_warp_sm        = mnc_roi_warp        (+ d_sm of d_out_rhwc as [R][PH*PW*C])
 *   mnc_maxpool2_rhwc_sm   = mnc_maxpool2_rhwc   (+ d_sm of d_out)
 *   mnc_mask_pool_sm       = mnc_mask_pool       (+ d_sm of d_out)
 * C%64==0 (fmt 1) / C%32==0 (fmt 2) so that an 8-channel group never straddles a stage. */
MNC_API int mnc_fc_pack_act(mnc_ctx* ctx, const float* d_a, void* d_a_sm, int M, int K, int f16);
/* Round 6.  The inverse: fp32 rows [M][K] out of a stage-major tensor (fmt 1 = fp16: the rounded values; fmt 2 = split bf16: hi + lo)
 * -- for a consumer, or the host, that needs the rows of a tensor whose producer wrote the stage-major form ONLY
 * (mnc_roi_warp_sm / mnc_box_mask_pool_ex with null fp32 outputs; mnc_amd/engine.py materialises such blobs on demand). */
MNC_API int mnc_fc_unpack_act(mnc_ctx* ctx, const void* d_a_sm, float* d_a, int M, int K, int fmt);
/* *ok = 1 when mnc_roi_warp_sm on this context (its layer conventions, its kernel choice for C channels / pool2) accepts
 * d_out_rhwc = NULL next to a stage-major output. */
MNC_API int mnc_roi_warp_sm_only_ok(mnc_ctx* ctx, int C, int pool2, int* ok);
MNC_API int mnc_fc_f16_pre(mnc_ctx* ctx, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias,
                           float* d_out, int M, int N, int K, int ldc, int act);
MNC_API int mnc_fc_bf16x3_pre(mnc_ctx* ctx, const void* d_a_sm, int m_stride, const void* d_w_packed, const float* d_bias,
                              float* d_out, int M, int N, int K, int ldc, int act);
/* The general form of the two: activations as fp32 rows (d_a) or stage-major (d_a_sm, m_stride) -- exactly one non-NULL -- and,
 * optionally (d_out_sm != NULL), the result rows written a second time in the stage-major form of the NEXT reduced-precision
 * InnerProduct (out_sm_fmt 1: [N/64][M][64] halves, N%64==0; 2: split bf16, N%32==0) by the K-split reduction: fc6 -> fc7 without a
 * conversion pass.  Bit for bit mnc_fc_pack_act of d_out. */
MNC_API int mnc_fc_f16_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed,
                          const float* d_bias, float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm, int out_sm_fmt);
MNC_API int mnc_fc_bf16x3_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed,
                             const float* d_bias, float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm,
                             int out_sm_fmt);
/* Round 6: the plain bf16 mode's stage-major form, FORMAT 3 = format 1's layout ([K/64][M][64] 2-byte values) with the values
 * rounded to bf16 (nearest even) instead of fp16.  Written by the same producers (sm_fmt = 3: mnc_roi_warp_sm, mnc_maxpool2_rhwc_sm,
 * mnc_mask_pool_sm, mnc_box_mask_pool[_ex] -- whose stage-major input may be format 3 too, with format-3 outputs -- and the K-split
 * reduction of an InnerProduct, out_sm_fmt = 3), by mnc_fc_pack_act(.., f16 = 2), read by mnc_fc_bf16_ex / mnc_fc_lowp_pair(mode 2)
 * and mnc_fc_unpack_act(.., fmt = 3). */
MNC_API int mnc_fc_bf16_ex(mnc_ctx* ctx, const float* d_a, const void* d_a_sm, int m_stride, const void* d_w_packed,
                           const float* d_bias, float* d_out, int M, int N, int K, int ldc, int act, void* d_out_sm, int out_sm_fmt);
MNC_API int mnc_roi_warp_sm(mnc_ctx* ctx, const float* d_feat_c8, int C, int H, int W, const float* d_rois, int R, int PH, int PW,
                            float spatial_scale, int pool2, float* d_out_rhwc, void* d_sm, int sm_fmt);
MNC_API int mnc_maxpool2_rhwc_sm(mnc_ctx* ctx, const float* d_in, float* d_out, int R, int PH, int PW, int C, void* d_sm,
                                 int sm_fmt);
MNC_API int mnc_mask_pool_sm(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_out, int R, int PH, int PW, int C,
                             int pool2, void* d_sm, int sm_fmt);
/* The box-feature Pooling (MAX 2x2/2 of the per-RoI tensor, test.prototxt:571-582) and MaskPooling + its Pooling (:631-650) of
 * the SAME tensor in one pass: d_box_out = mnc_maxpool2_rhwc(d_feat), d_mask_out = mnc_mask_pool(d_feat, d_mask, pool2 = 1), bit
 * for bit; d_feat is read once instead of twice.  d_box_sm / d_mask_sm: their stage-major second outputs (both or neither). */
MNC_API int mnc_box_mask_pool(mnc_ctx* ctx, const float* d_feat, const float* d_mask, float* d_box_out, float* d_mask_out, int R,
                              int PH, int PW, int C, void* d_box_sm, void* d_mask_sm, int sm_fmt);
/* Round 6.  The same pass with (a) the fp32 outputs optional -- both null when only the stage-major ones are consumed (60 of 210 MB
 * per call at 300 RoIs x 512 channels) -- and (b) the 14x14 tensor read from its stage-major fp16 form d_feat_sm (feat_sm_fmt = 1:
 * [PH*PW*C/64][R][64] halves, what mnc_roi_warp_sm wrote for fc6_maskest) when stage-major outputs are written: 60 instead of 120 MB
 * read, and the producer need not write the fp32 tensor (mnc_roi_warp_sm accepts d_out_rhwc = NULL with a stage-major output on the
 * SPEC's convention and the launcher's own kernel choice).  d_box_sm is then the same bits as from the fp32 tensor (rounding is monotonic); d_mask_sm
 * multiplies the ROUNDED features by the mask and differs in the last fp16 bit -- every executor of a graph must pass the same
 * inputs (csrc/pipeline.hip and engine.py both pass d_feat_sm whenever the producer wrote it).  feat_sm_fmt = 2: the split-bf16 form
 * ([PH*PW*C/32][R][4][hi x8 | lo x8]; a value is hi + lo, 16 significant bits -- the box pool too may then differ in the last fp16
 * bit).  feat_sm_fmt 0, or no stage-major outputs: d_feat is read as in mnc_box_mask_pool. */
MNC_API int mnc_box_mask_pool_ex(mnc_ctx* ctx, const float* d_feat, const void* d_feat_sm, int feat_sm_fmt, const float* d_mask,
                                 float* d_box_out, float* d_mask_out, int R, int PH, int PW, int C, void* d_box_sm, void* d_mask_sm,
                                 int sm_fmt);
/* Softmax over the last axis of [M][N] (test.prototxt cls_prob / seg_cls_prob). */
MNC_API int mnc_softmax_rows(mnc_ctx* ctx, const float* d_in, float* d_out, int M, int N);
/* Same with a row stride on the input (the input may be a column slice of a merged-GEMM output). */
MNC_API int mnc_softmax_rows_ld(mnc_ctx* ctx, const float* d_in, int ld_in, float* d_out, int M, int N);
/* Stand-alone ReLU (op 1) / Sigmoid (op 2) for graphs where the activation is not fused into its producer
 * (test.prototxt:540-545 `mask_output` when run unfused).  In-place allowed. */
MNC_API int mnc_eltwise(mnc_ctx* ctx, const float* d_in, float* d_out, size_t count, int op);
/* Strided 2-D device copy of float rows (Concat, test.prototxt:700-709, when the producers could not write in place). */
MNC_API int mnc_copy2d(mnc_ctx* ctx, float* d_dst, int dst_ld, const float* d_src, int src_ld, int rows, int cols);

/* ---------------------------------------------------------------------------------------------------------------
 * Device-resident forms of the three inference-time Python layers (the Python classes in mnc_amd/lib/pylayer remain the
 * API and the path for user-defined layers; the engine substitutes these for the stock classes).
 * ------------------------------------------------------------------------------------------------------------- */
/* ProposalLayer.forward (lib/pylayer/proposal_layer.py:52-175): d_cls_prob [2A][H][W], d_bbox_pred [4A][H][W] (NCHW,
 * batch 1), anchors_host [A][4] (transform.anchors.generate_anchors as float32), im_info = (im_h, im_w, im_scale).
 * Writes d_rois [post_nms_topn][5] (rows >= the row count are zero) and returns the row count in *num_rois_host (one 4-byte
 * D2H + stream sync).  num_rois_host == NULL: fully asynchronous, the count stays on the device until mnc_proposal_count
 * (the engine launches the heads on all post_nms_topn rows meanwhile and checks the count with the outputs).
 * Candidate order is score descending, anchor index ascending (the reference leaves tie order to numpy's sort). */
MNC_API int mnc_proposal(mnc_ctx* ctx, const float* d_cls_prob, const float* d_bbox_pred, int A, int H, int W,
                         const float* anchors_host, int feat_stride, float im_h, float im_w, float im_scale,
                         int pre_nms_topn, int post_nms_topn, float nms_thresh, float min_size, float* d_rois,
                         int* num_rois_host);
/* gpu_mask_voting (mnc_mask_voting above) with the inputs already on the device -- the engine's own outputs, produced on ctx's stream (no host round trip
 * between net.forward and the voting): d_boxes [n][4], d_masks [n][S][S], d_scores [n][num_classes]; outputs are host
 * arrays as above.  The library orders each class itself (order = NULL semantics). */
MNC_API int mnc_mask_voting_dev(mnc_ctx* ctx, const float* d_boxes, const float* d_masks, const float* d_scores, int n,
                                int num_classes, int mask_size, int max_per_image, float nms_thresh, float iou_thresh,
                                int image_height, int image_width, float* out_mask, int* out_box, float* out_score,
                                int* class_count, int* result_num);
/* gpu_mask_voting with inputs AND outputs on the device, fully asynchronous on ctx's stream (no host decision, no
 * synchronisation): the whole-image path's last stage, and the block the multi-GPU path gathers.
 *   d_records [record_cap][6 + S*S] float32: (x1, y1, x2, y2, score, class id 1..num_classes-1, S*S mask values) of the
 *             result rows in the reference's order (class-major, keep order); rows past the result count are zero (class 0).
 *   d_counts  [num_classes] int32: [0] = R, the number of result rows (> max_per_image only when scores tie at the global
 *             threshold; rows >= record_cap are not written), [c] = rows of class c.
 * The global threshold and the result rows (mask_transform.py:242-258) are chosen by a kernel (np.sort()[::-1] order, NaN
 * first).  Limits: n <= 4096, (num_classes-1) * min(max_per_image, n) <= 8192.  Records are bit-identical to mnc_mask_voting. */
MNC_API int mnc_vote_instances(mnc_ctx* ctx, const float* d_boxes, const float* d_masks, const float* d_scores, int n,
                               int num_classes, int mask_size, int max_per_image, float nms_thresh, float iou_thresh,
                               int image_height, int image_width, float* d_records, int record_cap, int* d_counts);
/* The tail of im_detect on the device (tools/demo.py:84-100, lib/caffeWrapper/TesterWrapper.py:240-260): d_boxes
 * [R1+R2][4] = clip(rois[:, 1:5] / scale, image) of stage-1 rois followed by stage-2 rois (float32 division, clamp to
 * [0, W-1] x [0, H-1] as transform/bbox_transform.py:clip_boxes). */
MNC_API int mnc_detect_tail(mnc_ctx* ctx, const float* d_rois1, int R1, const float* d_rois2, int R2, float scale,
                            int image_height, int image_width, float* d_boxes);

/* Row count of the last mnc_proposal on this context (4-byte D2H + stream sync). */
MNC_API int mnc_proposal_count(mnc_ctx* ctx, int* num_rois_host);
/* The sorted pre-NMS candidates of the last mnc_proposal on this context (parity tests teacher-force the NMS with them):
 * boxes_host [n][4], scores_host [n], *n_host = n.  Pass null arrays to query n only. */
MNC_API int mnc_proposal_candidates(mnc_ctx* ctx, float* boxes_host, float* scores_host, int capacity, int* n_host);
/* StageBridgeLayer.forward_test (lib/pylayer/stage_bridge_layer.py:237-255): per RoI the box regressor of the arg-max
 * class of d_probs (first maximum, background included) is applied and clipped to (im_h, im_w).  d_bbox_pred / d_probs
 * may be column slices (row strides ld_bbox / ld_probs). */
MNC_API int mnc_stage_bridge(mnc_ctx* ctx, const float* d_rois, const float* d_bbox_pred, int ld_bbox, const float* d_probs,
                             int ld_probs, int R, int K, float im_h, float im_w, float* d_rois_ext);

/* ---------------------------------------------------------------------------------------------------------------
 * The whole image in ONE call (mnc_load_weights / mnc_forward_image): what tools/demo.py does per image
 * (prepare_mnc_args :54-76, net.forward :79-83, the tail of im_detect :84-100, gpu_mask_voting :147) for the graph
 * models/VGG16/mnc_5stage/test.prototxt, as a native object -- no Python, no prototxt parser: the layer sequence of that file
 * (13 conv3x3 + 4 pools, RPN head + ProposalLayer, two head stages with the shared parameters, StageBridge) is fixed in
 * csrc/pipeline.hip with the fused plan the Python engine derives from the prototxt (warp+pool, FC+activation, Concat-free
 * column slices, merged sibling heads), the widths are configuration.  The launch sequence of an image size is captured in a
 * HIP graph on first use and replayed afterwards (use_graph); everything is asynchronous on the context's stream and the
 * call returns after ONE synchronisation, with the final instance records in host memory.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mnc_net mnc_net;
typedef struct mnc_net_config {
  int trunk_channels[5];   /* conv1_x .. conv5_x widths (64, 128, 256, 512, 512)                      test.prototxt:19-387 */
  int rpn_channels;        /* rpn_conv_3x3 width (512)                                                 :391-412 */
  int num_anchors;         /* 9; anchors[a*4 + k] = transform.anchors.generate_anchors() as float32   lib/transform/anchors.py:38-49 */
  float anchors[64];
  int feat_stride;         /* 16 */
  int pre_nms_topn, post_nms_topn;      /* 6000, 300                              lib/mnc_config.py TEST.RPN_{PRE,POST}_NMS_TOP_N */
  float rpn_nms_thresh, rpn_min_size;   /* 0.7, 16 */
  int mask_fc, mask_size;  /* fc6_maskest width 256, mask side 21 (mask_pred = mask_size^2 outputs)   :509-545 */
  int fc_dim;              /* fc6 / fc7 / fc6_mask / fc7_mask width 4096                               :584-696 */
  int num_classes;         /* 21 */
  int roi_size;            /* 14: ROIWarping 28x28 + MAX 2x2 in stage 2, 14x14 direct in stage 4      :479-505, 809-820 */
  float spatial_scale;     /* 0.0625 */
  int target_size, max_size;            /* 600, 1000: TEST.SCALES[0], TRAIN.MAX_SIZE (tools/demo.py:59) */
  double pixel_means[3];   /* cfg.PIXEL_MEANS, BGR */
  int max_per_image;       /* 100 */
  float vote_nms_thresh, vote_iou_thresh;   /* TEST.MASK_MERGE_NMS_THRESH 0.3, TEST.MASK_MERGE_IOU_THRESH 0.5 */
  int math;                /* 0 fp32, 1 bf16x3, 2 f16, 3 mixed = convolutions bf16x3 (fp32-class) + large InnerProducts fp16: the
                            * reduced-precision mode that keeps the 1e-3 bar; 4 bf16 = plain bf16, one product per term (BASELINE
                            * configs[2] as written; measured, outside the 1e-3 bar) (the engine's math modes) */
  int use_graph;           /* 1: replay a captured HIP graph per image size; 0: launch every kernel every time */
  int winograd;            /* fp32 math, the 3x3 convolutions: 4 = Winograd F(4x4,3x3) (mnc_conv3x3_wino4; default), 2 (or 1) =
                            * F(2x2,3x3) (mnc_conv3x3_wino), 0 = direct implicit GEMM */
  mnc_layer_conventions conventions;   /* ROIWarping / MaskResize / MaskPooling conventions.  The RoI kernels read them from the
                                        * CONTEXT: with conventions.inherit == 1 (mnc_net_default_config) mnc_net_create leaves the
                                        * context's alone (the SPEC on a fresh context, or what the host set with
                                        * mnc_ctx_set_layer_conventions); with inherit == 0 it applies this member to `ctx` (all
                                        * other fields zero = oracle/SPEC.md) -- for every net on that context */
} mnc_net_config;

/* The reference's values for every field (VGG-16 widths, lib/mnc_config.py defaults). */
MNC_API int mnc_net_default_config(mnc_net_config* cfg);
MNC_API int mnc_net_create(mnc_ctx* ctx, const mnc_net_config* cfg, mnc_net** out);
/* A second net over the SAME device weights as `parent` (its configuration too): own context `ctx` (same device: own stream,
 * scratch arenas, activation buffers, HIP graph), no weights of its own -- the reference shares parameters by `param { name }`
 * inside one net (test.prototxt:514-515 <-> :829-834); several images in flight on one GPU share them across nets the same
 * way (one 1.13 GB set instead of one per image in flight).  Packs `parent`'s weights first if that has not happened yet (its
 * parameters must all be set).  `parent` must outlive the nets that share with it: mnc_net_destroy(parent) fails with
 * MNC_ERR_STATE while one exists. */
MNC_API int mnc_net_create_shared(mnc_ctx* ctx, mnc_net* parent, mnc_net** out);
/* One parameter blob of one layer, in Caffe's own layout (what net.params[layer][index].data holds: Convolution
 * [Cout][Cin][3][3], InnerProduct [N][K] with K in (c,h,w) order, bias [N]).  Layers: conv1_1 .. conv5_3, rpn_conv_3x3,
 * rpn_cls_score, rpn_bbox_pred, fc6_maskest, mask_pred, fc6, fc7, fc6_mask, fc7_mask, cls_score, seg_cls_score, bbox_pred
 * (the *_ext layers share these by `param { name }`, test.prototxt:514-515 <-> :829-834).  index 0 = weights, 1 = bias. */
MNC_API int mnc_net_set_param(mnc_net* net, const char* layer, int index, const float* data_host, size_t count);
/* mnc_load_weights: every blob from a flat little-endian file written by mnc_amd.caffemodel.save_flat / tools/convert_weights.py:
 * "MNCW0001", uint32 n, then n x { uint16 name_len, name, uint8 blob index, uint8 ndim, uint32 dims[ndim], float32 data }.
 * Entries with a blob index > 1 (a third blob of a layer, e.g. BatchNorm's moving-average factor) are read over and ignored. */
MNC_API int mnc_net_load_file(mnc_net* net, const char* path);
/* One image.  bgr_host: uint8 [H][W][3] (BGR, as cv2.imread gives the reference).  records_host: [record_cap][6 + S*S] float32
 * = (x1, y1, x2, y2, score, class id, mask) of the voted instances, rows past the count zero; counts_host [num_classes]:
 * [0] = number of instances, [c] = instances of class c.  record_cap <= (num_classes-1) * max_per_image. */
MNC_API int mnc_forward_image(mnc_net* net, const unsigned char* bgr_host, int H, int W, float* records_host, int record_cap,
                              int* counts_host);
/* The two halves of mnc_forward_image, for hosts that keep several images in flight (one mnc_net + context + stream per image
 * in flight; independent images overlap on the GPU, so the latency-bound stretches of one -- proposal top-k, NMS scan, voting --
 * run beside the next one's convolutions):
 *   mnc_forward_image_async  stages the image and enqueues everything (graph replay or direct launches); no synchronisation.
 *                            *d_records / *d_counts (may be NULL) are the device-resident block mnc_gather_instances sends.
 *   mnc_net_fetch            waits for that image and hands out its records exactly as mnc_forward_image does. */
MNC_API int mnc_forward_image_async(mnc_net* net, const unsigned char* bgr_host, int H, int W, float** d_records, int** d_counts);
MNC_API int mnc_net_fetch(mnc_net* net, float* records_host, int record_cap, int* counts_host);
/* Device address and Caffe-order shape of an intermediate blob of the LAST image, for parity tests: "conv5_3" (c8),
 * "rpn_cls_prob_reshape", "rpn_bbox_pred", "rois", "rois_ext", "mask_proposal" [2R][S][S] (both stages stacked),
 * "seg_cls_prob" [2R][num_classes], "boxes" [2R][4], "head_scores" [2R][6*num_classes] = [cls_score | seg_cls_score | bbox_pred]
 * of both stages, "data" (the prepared network input), "records" (the instance block of mnc_forward_image_async).  dims
 * receives up to 4 ints, *ndim their number. */
MNC_API int mnc_net_blob(mnc_net* net, const char* name, void** d_ptr, int* dims, int* ndim);
MNC_API int mnc_net_destroy(mnc_net* net);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-GPU (the reference is single-GPU: batch is 1 per forward, lib/pylayer/proposal_layer.py:65).  Images are
 * sharded one per rank, one process and one context per GPU, weights replicated, no data-path collective.  The only exchange
 * is the all-gather of every rank's instance records (mnc_vote_instances' d_records, [100][447] float32 by default) issued as
 * ncclAllGather ON THE CONTEXT'S STREAM, device pointer to device pointer.  librccl is loaded at run time.
 *   mnc_comm_unique_id  rank 0 creates the 128-byte ncclUniqueId; the host program carries it to the other ranks.
 *   mnc_comm_init       ncclCommInitRank for this context's device (collective: every rank calls it).
 *   mnc_gather_instances d_recv [nranks][floats_per_rank] <- every rank's d_send [floats_per_rank]; asynchronous.
 * ------------------------------------------------------------------------------------------------------------- */
MNC_API int mnc_comm_unique_id(void* id_out, int capacity_bytes);
MNC_API int mnc_comm_init(mnc_ctx* ctx, const void* id, int nranks, int rank);
MNC_API int mnc_comm_info(mnc_ctx* ctx, int* nranks, int* rank, int* rccl_version);
MNC_API int mnc_gather_instances(mnc_ctx* ctx, const float* d_send, float* d_recv, size_t floats_per_rank);
MNC_API int mnc_comm_destroy(mnc_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MNC_HIP_H_ */
