// mnc_forward_image: the whole per-image hot path of MNC as one native call (include/mnc_hip.h, "The whole image in ONE call").
//
// What tools/demo.py does per image in the reference --
//     prepare_mnc_args (tools/demo.py:54-76)  ->  net.forward (:79-83)  ->  un-scale / clip / concat (:84-100)  ->
//     gpu_mask_voting (:147, lib/transform/mask_transform.py:213-286)
// -- for the graph models/VGG16/mnc_5stage/test.prototxt, with the layer sequence of that file written down here instead of
// parsed: this is the fixed-function form of the path, for hosts that are not Python (a cgo / JNI / plain C caller) and for the
// bench, where ~100 ctypes calls per image would otherwise sit next to a 3 ms GPU step.  The per-layer kernels are the same
// C-ABI entry points the Python engine calls (mnc_conv3x3*, mnc_fc*, mnc_roi_warp, mnc_proposal, mnc_vote_instances, ...), in
// the fused plan the engine derives from the prototxt, so the two executors produce the same bits (tests/test_gpu_pipeline.py).
//
// Launch model: every launch is asynchronous on the context's stream; the image goes up from a pinned staging buffer and the
// instance records come down into one, so the sequence [H2D, prep, trunk, RPN, proposal, stage 2/3 heads, bridge, stage 4/5
// heads, tail, voting, D2H] has no host dependency at all and is captured into a HIP graph the second time an image size is
// seen; later images of that size are one hipGraphLaunch + one stream synchronisation.  The RoI count of the ProposalLayer
// stays on the device (the heads run on all post_nms_topn rows, rows past the count are zero boxes); it comes down with the
// records, and only if fewer proposals survived are the heads re-run on the exact count -- what the reference computes.
#include <atomic>
#include <cmath>
#include <map>
#include <string>

#include "mnc_internal.h"

namespace mnc {
int* proposal_count_ptr(mnc_ctx* ctx);   // proposal.hip: device address of the last mnc_proposal's row count
}

using namespace mnc;

namespace {

struct HostBlob {
  std::vector<float> v;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

const char* kTrunk[13] = {"conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3",
                          "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"};
const int kTrunkStage[13] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};
const bool kPoolAfter[13] = {false, true, false, true, false, false, true, false, false, true, false, false, false};

inline int pool_out(int n) { return (n - 2 + 1) / 2 + 1; }       // Caffe MAX 2x2/2 pad 0, ceil

// cv2.resize INTER_LINEAR taps of one axis, as mnc_amd/prep.py:linear_taps (the definition shared with the numpy path):
// source coordinate (dst + 0.5) * (1/scale) - 0.5 in double, stored as float; floor; border clamps with a zero fraction.
void linear_taps(int n_dst, int n_src, double scale, int* lo, float* frac) {
  const double inv = 1.0 / scale;
  for (int i = 0; i < n_dst; ++i) {
    const float src = (float)(((double)i + 0.5) * inv - 0.5);
    long l = (long)std::floor((double)src);
    float f = (float)((double)src - (double)l);
    if (l < 0) { l = 0; f = 0.0f; }
    if (l >= n_src - 1) { l = n_src - 1; f = 0.0f; }
    lo[i] = (int)l;
    frac[i] = f;
  }
}

}  // namespace

struct mnc_net {
  mnc_ctx* ctx = nullptr;
  // Second context (own stream, own scratch arena) for the box-feature branch of a head stage, which does not depend on the
  // mask branch until the Concat: fc6 / fc7 run beside fc6_maskest .. fc7_mask, so each branch's small kernels (K-split
  // reductions, pools, mask resampling) and launch gaps sit under the other's GEMMs.  Not used while per-kernel events are
  // recorded.  Opt-in (MNC_BRANCH_STREAMS=1): measured +1 % for one image at a time (170.9 vs 169.3 images/s; two GEMMs that
  // each fill the chip gain nothing from sharing it) and -12 % with two images in flight (four streams: 172 vs 195), and the
  // overlap makes a kernel trace of the run disagree with the event-step durations.
  mnc_ctx* ctx_b = nullptr;
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
  mnc_net_config cfg;
  std::map<std::string, HostBlob> params;      // "<layer>/<index>" as given by the caller (Caffe layout)
  bool finalized = false;
  // mnc_net_create_shared: the device weights below belong to `owner` (this net holds copies of the pointers and frees none of
  // them); `sharers` counts the nets that borrow from this one
  mnc_net* owner = nullptr;
  std::atomic<int> sharers{0};
  // device weights
  float* w_c3 = nullptr;                       // conv1_1 [Cout][3][3][3]
  void* w_conv[14] = {nullptr};                // packed conv3x3 weights: trunk 1..12, [13] = rpn_conv_3x3
  bool packed_trunk = false;                   // bf16x3 / f16 / bf16 with every trunk layer on the tuned kernel: 2-byte activations
  bool conv_fast[14] = {false};                // tuned 3x3 kernels (Cout % 32 == 0) or the general convolution (reduced widths)
  float* b_conv[14] = {nullptr};               // biases: trunk 0..12 -> [0..12], rpn -> [13]
  float *w_rpn = nullptr, *b_rpn = nullptr;    // rpn_cls_score (2A rows) and rpn_bbox_pred (4A rows) as ONE [6A][RC] 1x1 convolution
  struct Fc { void* w = nullptr; float* b = nullptr; int N = 0, K = 0, kind = 0; };   // kind 0 fp32, 1 bf16x3, 2 f16, 3 plain bf16
  Fc fc_maskest, fc_maskpred, fc6, fc7, fc6m, fc7m, fc_heads;
  // geometry of the buffers below
  int cap_ph = 0, cap_pw = 0, cap_src = 0;
  DevBuf img, taps, data, act[13], pooled[4], act12_pk, rpn_out, rpn_score, rpn_prob;   // act12_pk: conv5_3 packed (for rpn_conv); rpn_score: [2A score | 4A bbox] planes
  float* rpn_bbox_p = nullptr;                                                      // = rpn_score + 2A planes (set per image size)
  DevBuf rois, rois_ext, feat14, h_mask, m14, box7, mask7, f6, f6m, join, heads, boxes, masks, scores;
  // the result block: [per-class counts: 256 B | the ProposalLayer's row count: 256 B | instance records] -- laid out like the
  // pinned buffer it is copied into, so that an image's results come down in ONE copy
  DevBuf outblk;
  DevBuf hwc5;            // conv5_3 pixel-major: made once per image, gathered from by both stages' ROIWarping
  float* records() const { return (float*)((char*)outblk.p + 512); }
  int* counts() const { return (int*)outblk.p; }
  int* prop_count() const { return (int*)((char*)outblk.p + 256); }
  // the per-RoI tensors a second time, in the stage-major 2-byte form their reduced-precision InnerProduct multiplies from
  // (written by the producing kernel's epilogue: mnc_roi_warp_sm / mnc_maxpool2_rhwc_sm / mnc_mask_pool_sm), bf16x3 / f16 modes
  DevBuf feat14_sm, box7_sm, mask7_sm, f6_sm, f6m_sm;       // f6 / f6m: written by fc6 / fc6_mask's reduction for fc7 / fc7_mask
  unsigned char* pin_img = nullptr; size_t pin_img_cap = 0;
  float* pin_out = nullptr; size_t pin_out_cap = 0;       // [counts (64 ints) | proposal count (64 ints) | records]
  // per-image state
  int H = 0, W = 0, OH = 0, OW = 0, fh = 0, fw = 0;
  float im_scale = 1.0f;
  int tap_key_h = -1, tap_key_w = -1;
  int last_r1 = 0, last_r2 = 0;
  bool fc_sm = true, fuse_pools = true;   // plan switches, read from the context's tuning values when the net is created
  bool fuse_small = true;                 // FUSE_SMALL (default on): heads_finish / rpn_heads instead of their separate launches
  bool tail_done = false;                 // per image: the stage-5 finish kernel has run im_detect's tail
  bool in_flight = false;      // an image has been launched and not fetched: its staging buffers are still in use
  // HIP graph of one image size
  hipGraphExec_t gexec = nullptr;
  int graph_h = -1, graph_w = -1, seen_h = -1, seen_w = -1;
  unsigned long graph_gen = 0;   // arena generation (mnc_ctx::arena_gen of both contexts) the graph was captured under
};

namespace {

int dev_ensure(mnc_net* n, DevBuf* b, size_t bytes) {
  if (bytes <= b->cap) return MNC_OK;
  if (b->p) {
    int rc = mnc_dev_free(n->ctx, b->p);
    if (rc) return rc;
    b->p = nullptr; b->cap = 0;
  }
  const size_t want = bytes + (bytes >> 3) + 256;
  int rc = mnc_dev_alloc(n->ctx, want, &b->p);
  if (rc) return rc;
  b->cap = want;
  if (n->gexec) { (void)hipGraphExecDestroy(n->gexec); n->gexec = nullptr; n->graph_h = n->graph_w = -1; }   // addresses changed
  return MNC_OK;
}

const HostBlob* find(mnc_net* n, const std::string& layer, int index) {
  auto it = n->params.find(layer + "/" + std::to_string(index));
  return it == n->params.end() ? nullptr : &it->second;
}

int upload(mnc_net* n, const std::vector<float>& v, float** out) {
  void* p = nullptr;
  int rc = mnc_dev_alloc(n->ctx, v.size() * 4, &p);
  if (rc) return rc;
  rc = mnc_h2d(n->ctx, p, v.data(), v.size() * 4);
  if (rc) return rc;
  *out = (float*)p;
  return MNC_OK;
}

#define NET_TRY(expr)        \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

int need(mnc_net* n, const char* layer, int index, size_t count, const HostBlob** out) {
  const HostBlob* b = find(n, layer, index);
  if (!b) {
    set_error("mnc_net: parameter %s/%d has not been set", layer, index);
    return MNC_ERR_STATE;
  }
  if (b->v.size() != count) {
    set_error("mnc_net: parameter %s/%d holds %zu values, the configured graph needs %zu", layer, index, b->v.size(), count);
    return MNC_ERR_INVALID;
  }
  *out = b;
  return MNC_OK;
}

// math 3 ("mixed", round 4): the convolutions in bf16x3 (fp32-class), the large InnerProducts in fp16 -- the reduced-precision mode
// that keeps the 1e-3 bar (mnc_hip.h, mnc_net_config::math)
// math 4 ("bf16"): plain bf16, one product per term, fp32 tensors between the layers (BASELINE configs[2] as written; measured only)
static inline int conv_math(const mnc_net_config& c) { return c.math == 3 ? 1 : c.math; }
static inline int lowp_mode(int cm) { return cm == 1 ? 0 : cm == 2 ? 1 : 2; }      // conv_math 1 / 2 / 4 -> mnc_conv3x3_lowp's mode
static inline int fc_math(const mnc_net_config& c) { return c.math == 3 ? 2 : c.math; }

// One InnerProduct's weights: [N][K] in Caffe order; geo = (C, PH, PW) when the bottom is a per-RoI feature (the engine's
// rows are (h, w, c): permute the columns once), then the math mode's packed form when the product is large enough
// (engine.py: 2*M*N*K >= 2e9 with M = post_nms_topn; f16 needs K % 64 == 0, bf16x3 K % 32 == 0).
int prepare_fc(mnc_net* n, const char* layer, int N, int K, int C, int PH, int PW, mnc_net::Fc* fc) {
  const HostBlob *w, *b;
  NET_TRY(need(n, layer, 0, (size_t)N * K, &w));
  NET_TRY(need(n, layer, 1, (size_t)N, &b));
  fc->N = N; fc->K = K;
  NET_TRY(upload(n, b->v, &fc->b));
  float* raw = nullptr;
  NET_TRY(upload(n, w->v, &raw));
  if (PH * PW > 1) {
    void* perm = nullptr;
    NET_TRY(mnc_dev_alloc(n->ctx, (size_t)N * K * 4, &perm));
    NET_TRY(mnc_pack_fc_weights(n->ctx, raw, (float*)perm, N, C, PH, PW));
    NET_TRY(mnc_dev_free(n->ctx, raw));
    raw = (float*)perm;
  }
  const bool big = 2.0 * n->cfg.post_nms_topn * (double)N * (double)K >= 2.0e9;
  // mixed: an InnerProduct over more than 50 000 inputs (fc6_maskest: 100 352) stays split-bf16 -- fp16 operands leave the sigmoid
  // masks behind it at 1.0-1.2e-3, the only head output of the f16 mode on the wrong side of the bar (engine.py: the same rule)
  const int fm = (n->cfg.math == 3 && K > 50000) ? 1 : fc_math(n->cfg);
  const bool bf = fm == 4 && K % 64 == 0 && big;
  const bool f16 = (fm == 2 && K % 64 == 0 && big) || bf;
  const bool x3 = !f16 && (fm == 1 || fm == 2) && K % 32 == 0 && big;
  fc->kind = bf ? 3 : f16 ? 2 : x3 ? 1 : 0;
  if (fc->kind == 0) { fc->w = raw; return MNC_OK; }
  void* packed = nullptr;
  NET_TRY(mnc_dev_alloc(n->ctx, (size_t)((N + 127) / 128) * 128 * K * (f16 ? 2 : 4), &packed));
  NET_TRY(bf ? mnc_pack_fc_bf16(n->ctx, raw, packed, N, K)
             : f16 ? mnc_pack_fc_f16(n->ctx, raw, packed, N, K) : mnc_pack_fc_bf16x3(n->ctx, raw, packed, N, K));
  NET_TRY(mnc_dev_free(n->ctx, raw));
  fc->w = packed;
  return MNC_OK;
}

int run_fc(mnc_ctx* ctx, const mnc_net::Fc& fc, const float* a, float* out, int M, int ldc, int act) {
  if (M == 0) return MNC_OK;
  if (fc.kind == 3) return mnc_fc_bf16(ctx, a, fc.w, fc.b, out, M, fc.N, fc.K, ldc, act);
  if (fc.kind == 2) return mnc_fc_f16(ctx, a, fc.w, fc.b, out, M, fc.N, fc.K, ldc, act);
  if (fc.kind == 1) return mnc_fc_bf16x3(ctx, a, fc.w, fc.b, out, M, fc.N, fc.K, ldc, act);
  return mnc_fc(ctx, a, (const float*)fc.w, fc.b, out, M, fc.N, fc.K, ldc, act);
}
int run_fc(mnc_net* n, const mnc_net::Fc& fc, const float* a, float* out, int M, int ldc, int act) {
  return run_fc(n->ctx, fc, a, out, M, ldc, act);
}
// 1 (fp16) / 2 (split bf16): the stage-major activation form this InnerProduct's kernel multiplies from, when the producer of its
// per-RoI input (C channels per position) can write it (an 8-channel group must not straddle a stage); 0: fp32 rows only.
// FC_SM=0 (mnc_ctx_set_tuning) switches the second outputs off (the InnerProduct then converts the fp32 rows itself, as in round 1).
int sm_format(const mnc_net* n, const mnc_net::Fc& fc, int C) {
  if (!n->fc_sm) return 0;
  if (fc.kind == 2) return C % 64 == 0 ? 1 : 0;
  if (fc.kind == 1) return C % 32 == 0 ? 2 : 0;
  if (fc.kind == 3) return C % 64 == 0 ? 3 : 0;          // plain bf16 (round 6): fp16's layout, bf16 values
  return 0;
}
// the InnerProduct on rows that exist in both forms: a_sm (m_stride = M rows) when the kernel takes it, the fp32 rows otherwise
// out_sm / out_fmt: the result rows a second time in the form the NEXT InnerProduct (of format out_fmt) multiplies from
int run_fc_sm(mnc_ctx* ctx, const mnc_net::Fc& fc, const float* a, const void* a_sm, int fmt, float* out, int M, int ldc, int act,
              void* out_sm = nullptr, int out_fmt = 0) {
  if (M == 0) return MNC_OK;
  if (!out_sm) out_fmt = 0;
  if (fc.kind == 2) return mnc_fc_f16_ex(ctx, fmt == 1 ? nullptr : a, fmt == 1 ? a_sm : nullptr, M, fc.w, fc.b, out, M, fc.N, fc.K, ldc,
                                         act, out_fmt ? out_sm : nullptr, out_fmt);
  if (fc.kind == 1) return mnc_fc_bf16x3_ex(ctx, fmt == 2 ? nullptr : a, fmt == 2 ? a_sm : nullptr, M, fc.w, fc.b, out, M, fc.N, fc.K,
                                            ldc, act, out_fmt ? out_sm : nullptr, out_fmt);
  if (fc.kind == 3) return mnc_fc_bf16_ex(ctx, fmt == 3 ? nullptr : a, fmt == 3 ? a_sm : nullptr, M, fc.w, fc.b, out, M, fc.N, fc.K, ldc,
                                          act, out_fmt ? out_sm : nullptr, out_fmt);
  return run_fc(ctx, fc, a, out, M, ldc, act);
}

int finalize(mnc_net* n) {
  if (n->finalized) return MNC_OK;
  const mnc_net_config& c = n->cfg;
  const HostBlob *w, *b;
  // trunk + rpn_conv_3x3
  int cin = 3;
  for (int i = 0; i < 14; ++i) {
    const char* name = i < 13 ? kTrunk[i] : "rpn_conv_3x3";
    const int cout = i < 13 ? c.trunk_channels[kTrunkStage[i]] : c.rpn_channels;
    NET_TRY(need(n, name, 0, (size_t)cout * cin * 9, &w));
    NET_TRY(need(n, name, 1, (size_t)cout, &b));
    NET_TRY(upload(n, b->v, &n->b_conv[i]));
    if (i == 0) {
      NET_TRY(upload(n, w->v, &n->w_c3));
    } else {
      float* raw = nullptr;
      NET_TRY(upload(n, w->v, &raw));
      n->conv_fast[i] = cout % 32 == 0;        // engine.py:_conv_kind: 'fast3x3' needs Cout % 32 == 0, otherwise 'general'
      if (n->conv_fast[i]) {
        const int cm = conv_math(c);
        const int pitch = c.winograd == 4 ? 288 : c.winograd ? 136 : 76;
        const size_t wbytes = cm == 0 ? (size_t)(cin / 8) * cout * pitch * 4 : mnc_conv3x3_lowp_weight_bytes(lowp_mode(cm), cout, cin);
        NET_TRY(mnc_dev_alloc(n->ctx, wbytes, &n->w_conv[i]));
        NET_TRY(cm == 0 ? (c.winograd == 4 ? mnc_pack_conv3x3_wino4(n->ctx, raw, (float*)n->w_conv[i], cout, cin)
                               : c.winograd ? mnc_pack_conv3x3_wino(n->ctx, raw, (float*)n->w_conv[i], cout, cin)
                                            : mnc_pack_conv3x3_weights(n->ctx, raw, (float*)n->w_conv[i], cout, cin))
                : cm == 1 ? mnc_pack_conv3x3_bf16x3(n->ctx, raw, n->w_conv[i], cout, cin)
                : cm == 4 ? mnc_pack_conv3x3_bf16(n->ctx, raw, n->w_conv[i], cout, cin)
                              : mnc_pack_conv3x3_f16(n->ctx, raw, n->w_conv[i], cout, cin));
      } else if (conv_math(c) == 2) {
        NET_TRY(mnc_dev_alloc(n->ctx, (size_t)9 * ((cin + 31) / 32) * 32 * cout * 2, &n->w_conv[i]));
        NET_TRY(mnc_pack_conv_weights_f16(n->ctx, raw, n->w_conv[i], cout, cin, 3, 3));
      } else {
        NET_TRY(mnc_dev_alloc(n->ctx, (size_t)cout * cin * 9 * 4, &n->w_conv[i]));
        NET_TRY(mnc_pack_conv_weights(n->ctx, raw, (float*)n->w_conv[i], cout, cin, 3, 3));
      }
      NET_TRY(mnc_dev_free(n->ctx, raw));
    }
    cin = cout;
  }
  n->packed_trunk = conv_math(c) != 0 && tune(n->ctx, T_PACKED_ACT, 1) != 0;
  for (int i = 1; i < 13; ++i) n->packed_trunk = n->packed_trunk && n->conv_fast[i];
  const int A = c.num_anchors, RC = c.rpn_channels;
  {
    // the two sibling 1x1 heads read the same rpn_output: one launch over the concatenated weights, their NCHW outputs are the
    // first 2A and the last 4A planes of one buffer (same arithmetic per channel; one 22 us latency-bound launch instead of two)
    const HostBlob *wc, *bc, *wb, *bb;
    NET_TRY(need(n, "rpn_cls_score", 0, (size_t)2 * A * RC, &wc));
    NET_TRY(need(n, "rpn_cls_score", 1, (size_t)2 * A, &bc));
    NET_TRY(need(n, "rpn_bbox_pred", 0, (size_t)4 * A * RC, &wb));
    NET_TRY(need(n, "rpn_bbox_pred", 1, (size_t)4 * A, &bb));
    std::vector<float> wcat(wc->v), bcat(bc->v);
    wcat.insert(wcat.end(), wb->v.begin(), wb->v.end());
    bcat.insert(bcat.end(), bb->v.begin(), bb->v.end());
    NET_TRY(upload(n, wcat, &n->w_rpn));
    NET_TRY(upload(n, bcat, &n->b_rpn));
  }
  // heads (shared by both stages, test.prototxt:514-515 <-> :829-834)
  const int C5 = c.trunk_channels[4], P = c.roi_size, S = c.mask_size, F = c.fc_dim, K = c.num_classes;
  NET_TRY(prepare_fc(n, "fc6_maskest", c.mask_fc, C5 * P * P, C5, P, P, &n->fc_maskest));
  NET_TRY(prepare_fc(n, "mask_pred", S * S, c.mask_fc, c.mask_fc, 1, 1, &n->fc_maskpred));
  NET_TRY(prepare_fc(n, "fc6", F, C5 * (P / 2) * (P / 2), C5, P / 2, P / 2, &n->fc6));
  NET_TRY(prepare_fc(n, "fc7", F, F, F, 1, 1, &n->fc7));
  NET_TRY(prepare_fc(n, "fc6_mask", F, C5 * (P / 2) * (P / 2), C5, P / 2, P / 2, &n->fc6m));
  NET_TRY(prepare_fc(n, "fc7_mask", F, F, F, 1, 1, &n->fc7m));
  {   // cls_score | seg_cls_score | bbox_pred as ONE GEMM over the concatenated weights (sibling InnerProducts on join_box_mask)
    const char* names[3] = {"cls_score", "seg_cls_score", "bbox_pred"};
    const int widths[3] = {K, K, 4 * K};
    std::vector<float> W, B;
    for (int i = 0; i < 3; ++i) {
      NET_TRY(need(n, names[i], 0, (size_t)widths[i] * 2 * F, &w));
      NET_TRY(need(n, names[i], 1, (size_t)widths[i], &b));
      W.insert(W.end(), w->v.begin(), w->v.end());
      B.insert(B.end(), b->v.begin(), b->v.end());
    }
    n->fc_heads.N = 6 * K; n->fc_heads.K = 2 * F; n->fc_heads.kind = 0;
    float* dw = nullptr;
    NET_TRY(upload(n, W, &dw));
    n->fc_heads.w = dw;
    NET_TRY(upload(n, B, &n->fc_heads.b));
  }
  NET_TRY(mnc_ctx_sync(n->ctx));
  n->params.clear();                 // the host copies are no longer needed
  n->finalized = true;
  return MNC_OK;
}

// Buffers for an image whose network input is OH x OW; fixed-size RoI buffers on first use.
int ensure_buffers(mnc_net* n, int H, int W, int OH, int OW) {
  const mnc_net_config& c = n->cfg;
  NET_TRY(dev_ensure(n, &n->img, (size_t)H * W * 3));
  NET_TRY(dev_ensure(n, &n->taps, (size_t)(2 * OW + 2 * OH) * 4));
  NET_TRY(dev_ensure(n, &n->data, (size_t)3 * OH * OW * 4));
  int h = OH, w = OW, pi = 0;
  for (int i = 0; i < 13; ++i) {
    const int ch = c.trunk_channels[kTrunkStage[i]];
    NET_TRY(dev_ensure(n, &n->act[i], (size_t)ch * h * w * 4));
    if (kPoolAfter[i]) {
      h = pool_out(h); w = pool_out(w);
      NET_TRY(dev_ensure(n, &n->pooled[pi++], (size_t)ch * h * w * 4));
    }
  }
  const int A = c.num_anchors;
  if (n->packed_trunk) NET_TRY(dev_ensure(n, &n->act12_pk, (size_t)c.trunk_channels[4] * h * w * 4));
  NET_TRY(dev_ensure(n, &n->hwc5, (size_t)c.trunk_channels[4] * h * w * 4));
  NET_TRY(dev_ensure(n, &n->rpn_out, (size_t)c.rpn_channels * h * w * 4));
  NET_TRY(dev_ensure(n, &n->rpn_score, (size_t)6 * A * h * w * 4));
  NET_TRY(dev_ensure(n, &n->rpn_prob, (size_t)2 * A * h * w * 4));
  n->rpn_bbox_p = (float*)n->rpn_score.p + (size_t)2 * A * h * w;
  const int R = c.post_nms_topn, C5 = c.trunk_channels[4], P = c.roi_size, S = c.mask_size, F = c.fc_dim, K = c.num_classes;
  NET_TRY(dev_ensure(n, &n->rois, (size_t)R * 5 * 4));
  NET_TRY(dev_ensure(n, &n->rois_ext, (size_t)R * 5 * 4));
  NET_TRY(dev_ensure(n, &n->feat14, (size_t)R * P * P * C5 * 4));
  NET_TRY(dev_ensure(n, &n->h_mask, (size_t)R * c.mask_fc * 4));
  NET_TRY(dev_ensure(n, &n->m14, (size_t)R * P * P * 4));
  NET_TRY(dev_ensure(n, &n->box7, (size_t)R * (P / 2) * (P / 2) * C5 * 4));
  NET_TRY(dev_ensure(n, &n->mask7, (size_t)R * (P / 2) * (P / 2) * C5 * 4));
  if (sm_format(n, n->fc_maskest, C5)) NET_TRY(dev_ensure(n, &n->feat14_sm, (size_t)R * P * P * C5 * (n->fc_maskest.kind == 1 ? 4 : 2)));
  if (sm_format(n, n->fc6, C5)) NET_TRY(dev_ensure(n, &n->box7_sm, (size_t)R * (P / 2) * (P / 2) * C5 * (n->fc6.kind == 1 ? 4 : 2)));
  if (sm_format(n, n->fc6m, C5)) NET_TRY(dev_ensure(n, &n->mask7_sm, (size_t)R * (P / 2) * (P / 2) * C5 * (n->fc6m.kind == 1 ? 4 : 2)));
  NET_TRY(dev_ensure(n, &n->f6, (size_t)R * F * 4));
  NET_TRY(dev_ensure(n, &n->f6m, (size_t)R * F * 4));
  if (sm_format(n, n->fc7, F)) NET_TRY(dev_ensure(n, &n->f6_sm, (size_t)R * F * (n->fc7.kind == 1 ? 4 : 2)));
  if (sm_format(n, n->fc7m, F)) NET_TRY(dev_ensure(n, &n->f6m_sm, (size_t)R * F * (n->fc7m.kind == 1 ? 4 : 2)));
  NET_TRY(dev_ensure(n, &n->join, (size_t)R * 2 * F * 4));
  NET_TRY(dev_ensure(n, &n->heads, (size_t)2 * R * 6 * K * 4));        // both stages' rows (stage 4/5 behind stage 2/3)
  NET_TRY(dev_ensure(n, &n->boxes, (size_t)2 * R * 4 * 4));
  NET_TRY(dev_ensure(n, &n->masks, (size_t)2 * R * S * S * 4));
  NET_TRY(dev_ensure(n, &n->scores, (size_t)2 * R * K * 4));
  const size_t rows = (size_t)(K - 1) * c.max_per_image;
  NET_TRY(dev_ensure(n, &n->outblk, 512 + rows * (6 + S * S) * 4));
  if ((size_t)H * W * 3 > n->pin_img_cap) {
    if (n->pin_img) NET_TRY(mnc_host_free(n->ctx, n->pin_img));
    n->pin_img = nullptr;
    NET_TRY(mnc_host_alloc(n->ctx, (size_t)H * W * 3 + 4096, (void**)&n->pin_img));
    n->pin_img_cap = (size_t)H * W * 3 + 4096;
    if (n->gexec) { (void)hipGraphExecDestroy(n->gexec); n->gexec = nullptr; n->graph_h = n->graph_w = -1; }
  }
  const size_t out_bytes = 512 + rows * (6 + S * S) * 4;
  if (out_bytes > n->pin_out_cap) {
    if (n->pin_out) NET_TRY(mnc_host_free(n->ctx, n->pin_out));
    n->pin_out = nullptr;
    NET_TRY(mnc_host_alloc(n->ctx, out_bytes, (void**)&n->pin_out));
    n->pin_out_cap = out_bytes;
    if (n->gexec) { (void)hipGraphExecDestroy(n->gexec); n->gexec = nullptr; n->graph_h = n->graph_w = -1; }
  }
  return MNC_OK;
}

// prepare_mnc_args (tools/demo.py:54-76): scale, network-input size, tap tables (uploaded when the geometry changes).
int set_geometry(mnc_net* n, int H, int W) {
  const mnc_net_config& c = n->cfg;
  const int shortside = H < W ? H : W, longside = H < W ? W : H;
  double scale = (double)c.target_size / (double)shortside;
  if (std::nearbyint(scale * longside) > c.max_size) scale = (double)c.max_size / (double)longside;    // np.round: half to even
  const int OH = (int)std::nearbyint(H * scale), OW = (int)std::nearbyint(W * scale);                  // python round()
  NET_TRY(ensure_buffers(n, H, W, OH, OW));
  n->H = H; n->W = W; n->OH = OH; n->OW = OW;
  n->im_scale = (float)scale;
  if (n->tap_key_h != H || n->tap_key_w != W) {
    std::vector<int> lo(OW > OH ? OW : OH);
    std::vector<float> fr(lo.size());
    std::vector<float> table((size_t)2 * OW + 2 * OH);
    linear_taps(OW, W, scale, lo.data(), fr.data());
    memcpy(table.data(), lo.data(), (size_t)OW * 4);
    memcpy(table.data() + OW, fr.data(), (size_t)OW * 4);
    linear_taps(OH, H, scale, lo.data(), fr.data());
    memcpy(table.data() + 2 * OW, lo.data(), (size_t)OH * 4);
    memcpy(table.data() + 2 * OW + OH, fr.data(), (size_t)OH * 4);
    NET_TRY(mnc_h2d(n->ctx, n->taps.p, table.data(), table.size() * 4));
    n->tap_key_h = H; n->tap_key_w = W;
  }
  return MNC_OK;
}

int conv3(mnc_net* n, int i, const float* in, float* out, int h, int w, int cin, int cout) {
  const mnc_net_config& c = n->cfg;
  const int cm = conv_math(c);
  if (!n->conv_fast[i]) {
    if (cm == 2) return mnc_conv2d_f16(n->ctx, in, n->w_conv[i], n->b_conv[i], nullptr, out, h, w, cin, cout, 3, 3, 1, 1, 1);
    return mnc_conv2d(n->ctx, in, (const float*)n->w_conv[i], n->b_conv[i], nullptr, out, h, w, cin, cout, 3, 3, 1, 1, 1);
  }
  if (cm == 0 && c.winograd == 4)
    return mnc_conv3x3_wino4(n->ctx, in, (const float*)n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
  if (cm == 0 && c.winograd)
    return mnc_conv3x3_wino(n->ctx, in, (const float*)n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
  if (cm == 0) return mnc_conv3x3(n->ctx, in, (const float*)n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
  if (cm == 1) return mnc_conv3x3_bf16x3(n->ctx, in, n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
  if (cm == 4) return mnc_conv3x3_bf16(n->ctx, in, n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
  return mnc_conv3x3_f16(n->ctx, in, n->w_conv[i], n->b_conv[i], out, h, w, cin, cout, 1);
}

// data -> conv5_3 -> RPN -> rois (test.prototxt:19-475)
int run_trunk(mnc_net* n) {
  const mnc_net_config& c = n->cfg;
  mnc_ctx* ctx = n->ctx;
  const float* t = (const float*)n->taps.p;
  NET_TRY(mnc_prep_image(ctx, (const unsigned char*)n->img.p, n->H, n->W, c.pixel_means, (const int*)t, t + n->OW, n->OW,
                         (const int*)(t + 2 * n->OW), t + 2 * n->OW + n->OH, n->OH, (float*)n->data.p, n->OH, n->OW));
  int h = n->OH, w = n->OW, pi = 0, cin = 3;
  const float* cur = (const float*)n->data.p;
  if (n->packed_trunk) {
    // bf16x3 / f16 / bf16: 2-byte activation tensors between the MFMA layers (include/mnc_hip.h "Packed 2-byte activations"): conv1_1
    // and every convolution's epilogue write the form the next layer multiplies from; conv5_3 writes fp32 c8 for the RoI warps AND
    // the packed form for the RPN convolution.  Bit for bit the fp32-tensor route (test_gpu_ops.py::test_conv3x3_packed_activations).
    const int mode = lowp_mode(conv_math(c));
    const void* pc = cur;
    for (int i = 0; i < 13; ++i) {
      const int cout = c.trunk_channels[kTrunkStage[i]];
      void* out = n->act[i].p;
      if (kPoolAfter[i] && i > 0 && n->fuse_small) {
        // conv + ReLU + MAX 2x2/2 in one kernel (round 6): the full-resolution blob is not produced
        void* p = n->pooled[pi++].p;
        NET_TRY(mnc_conv3x3_lowp_pool(ctx, mode, pc, n->w_conv[i], n->b_conv[i], p, h, w, cin, cout, 1));
        h = pool_out(h); w = pool_out(w);
        pc = p; cin = cout;
        continue;
      }
      if (i == 0) NET_TRY(mnc_conv3x3_c3_fmt(ctx, cur, n->w_c3, n->b_conv[0], out, h, w, cout, 1, mode + 1));
      else if (i < 12) NET_TRY(mnc_conv3x3_lowp(ctx, mode, pc, n->w_conv[i], n->b_conv[i], out, nullptr, h, w, cin, cout, 1));
      else NET_TRY(mnc_conv3x3_lowp(ctx, mode, pc, n->w_conv[i], n->b_conv[i], n->act12_pk.p, (float*)out, h, w, cin, cout, 1));
      pc = out; cin = cout;
      if (kPoolAfter[i]) {
        void* p = n->pooled[pi++].p;
        NET_TRY(mode == 1 ? mnc_maxpool2_c8_f16(ctx, pc, p, cout, h, w)
                          : mode == 2 ? mnc_maxpool2_c8_bf16(ctx, pc, p, cout, h, w) : mnc_maxpool2_c8_bf16x3(ctx, pc, p, cout, h, w));
        h = pool_out(h); w = pool_out(w);
        pc = p;
      }
    }
    cur = (const float*)pc;
  } else
  for (int i = 0; i < 13; ++i) {
    const int cout = c.trunk_channels[kTrunkStage[i]];
    float* out = (float*)n->act[i].p;
    if (kPoolAfter[i] && i > 0 && conv_math(c) == 0 && c.winograd && n->conv_fast[i]) {
      // conv + ReLU + MAX 2x2/2 in one kernel (the engine's fused plan; the full-resolution blob is not produced)
      float* p = (float*)n->pooled[pi++].p;
      NET_TRY(c.winograd == 4 ? mnc_conv3x3_wino4_pool(ctx, cur, (const float*)n->w_conv[i], n->b_conv[i], p, h, w, cin, cout, 1)
                              : mnc_conv3x3_wino_pool(ctx, cur, (const float*)n->w_conv[i], n->b_conv[i], p, h, w, cin, cout, 1));
      h = pool_out(h); w = pool_out(w);
      cur = p; cin = cout;
      continue;
    }
    if (i == 0) NET_TRY(mnc_conv3x3_c3(ctx, cur, n->w_c3, n->b_conv[0], out, h, w, cout, 1));
    else NET_TRY(conv3(n, i, cur, out, h, w, cin, cout));
    cur = out; cin = cout;
    if (kPoolAfter[i]) {
      float* p = (float*)n->pooled[pi++].p;
      NET_TRY(mnc_maxpool2_c8(ctx, cur, p, cout, h, w));
      h = pool_out(h); w = pool_out(w);
      cur = p;
    }
  }
  n->fh = h; n->fw = w;
  if (n->fuse_small)      // (both head stages warp from this copy; otherwise each mnc_roi_warp_sm call makes its own)
    NET_TRY(c8_to_hwc_launch(ctx, (const float*)n->act[12].p, (float*)n->hwc5.p, c.trunk_channels[4], h, w));
  const int A = c.num_anchors;
  if (n->packed_trunk && n->conv_fast[13])
    NET_TRY(mnc_conv3x3_lowp(ctx, lowp_mode(conv_math(c)), n->act12_pk.p, n->w_conv[13], n->b_conv[13], nullptr, (float*)n->rpn_out.p, h, w,
                             cin, c.rpn_channels, 1));
  else NET_TRY(conv3(n, 13, cur, (float*)n->rpn_out.p, h, w, cin, c.rpn_channels));
  if (n->fuse_small) {
    NET_TRY(mnc_rpn_heads(ctx, (const float*)n->rpn_out.p, n->w_rpn, n->b_rpn, (float*)n->rpn_score.p, (float*)n->rpn_prob.p, h, w,
                          c.rpn_channels, A));
  } else {
    NET_TRY(mnc_conv1x1_to_nchw(ctx, (const float*)n->rpn_out.p, n->w_rpn, n->b_rpn, (float*)n->rpn_score.p, h, w, c.rpn_channels,
                                6 * A));
    NET_TRY(mnc_rpn_softmax(ctx, (const float*)n->rpn_score.p, (float*)n->rpn_prob.p, A, h, w));
  }
  NET_TRY(mnc_proposal(ctx, (const float*)n->rpn_prob.p, (const float*)n->rpn_bbox_p, A, h, w, c.anchors, c.feat_stride,
                       (float)n->OH, (float)n->OW, n->im_scale, c.pre_nms_topn, c.post_nms_topn, c.rpn_nms_thresh,
                       c.rpn_min_size, (float*)n->rois.p, nullptr));
  return MNC_OK;
}

// One head stage on R rois (test.prototxt:479-785 resp. :809-1106): mask head, box + mask feature branches, the three sibling
// classifiers; masks / seg scores land in rows [row0, row0 + R) of the stacked result arrays.
int run_stage(mnc_net* n, const float* rois, int R, bool second, int row0) {
  const mnc_net_config& c = n->cfg;
  mnc_ctx* ctx = n->ctx;
  const int C5 = c.trunk_channels[4], P = c.roi_size, S = c.mask_size, F = c.fc_dim, K = c.num_classes;
  const float* conv5 = (const float*)n->act[12].p;
  float* feat14 = (float*)n->feat14.p;
  // stage 2: ROIWarping 28x28 + MAX 2x2/2 fused; stage 4: ROIWarping 14x14 directly (test.prototxt:479-505 vs :809-820)
  const int sm_feat = sm_format(n, n->fc_maskest, C5), sm_box = sm_format(n, n->fc6, C5), sm_mask = sm_format(n, n->fc6m, C5);
  // Round 6: in the fp16 InnerProduct modes whose head runs as paired launches the 14x14 tensor exists in its stage-major fp16 form
  // only -- fc6_maskest multiplies from it, the poolings read it (mnc_box_mask_pool_ex) -- 120 MB per stage neither written nor read
  const bool fork0 = n->ctx_b && ctx->profiling == 0 && R > 0;
  const bool lowp_pairs = n->fuse_small && n->fuse_pools && sm_box == sm_mask && !fork0 && n->fc6.kind != 0 && n->fc6m.kind == n->fc6.kind &&
                          n->fc7.kind == n->fc6.kind && n->fc7m.kind == n->fc6.kind && n->fc6.K == n->fc6m.K &&
                          sm_format(n, n->fc7, F) == sm_format(n, n->fc7m, F) && R > 0;
  const bool sm_only = lowp_pairs && sm_box != 0;                 // box7 / mask7: stage-major only
  const bool feat_sm_only = sm_only && sm_feat != 0 && (P * P * C5) % 64 == 0 && roi_warp_sm_only_ok(ctx, C5, second ? 0 : 1);
  if (feat_sm_only) feat14 = nullptr;
  if (n->fuse_small)
    NET_TRY(roi_warp_from_hwc(ctx, (const float*)n->hwc5.p, C5, n->fh, n->fw, rois, R, P, P, c.spatial_scale, second ? 0 : 1, feat14,
                              n->feat14_sm.p, sm_feat));
  else
    NET_TRY(mnc_roi_warp_sm(ctx, conv5, C5, n->fh, n->fw, rois, R, P, P, c.spatial_scale, second ? 0 : 1, feat14, n->feat14_sm.p,
                            sm_feat));
  float* masks = (float*)n->masks.p + (size_t)row0 * S * S;
  NET_TRY(run_fc_sm(ctx, n->fc_maskest, feat14, n->feat14_sm.p, sm_feat, (float*)n->h_mask.p, R, c.mask_fc, 1));
  NET_TRY(run_fc(n, n->fc_maskpred, (const float*)n->h_mask.p, masks, R, S * S, 2));            // + Sigmoid; MaskLayer = reshape
  NET_TRY(mnc_mask_resize(ctx, masks, (float*)n->m14.p, R, S, S, P, P));
  float* join = (float*)n->join.p;                                                             // Concat(fc7_mask, fc7): column slices
  // box-feature Pooling and MaskPooling + Pooling read the same 14x14 tensor: one pass (mnc_box_mask_pool) when their
  // InnerProducts take the same activation form (always, with fc6 / fc6_mask of equal shape); MNC_FUSE_POOLS=0: two kernels
  const bool one_pass = n->fuse_pools && sm_box == sm_mask;
  // (round 6: when the paired reduced-precision InnerProducts below read the stage-major forms, the fp32 copies are not written)
  if (one_pass)
    NET_TRY(mnc_box_mask_pool_ex(ctx, feat14, sm_feat ? n->feat14_sm.p : nullptr, sm_feat, (const float*)n->m14.p,
                                 sm_only ? nullptr : (float*)n->box7.p, sm_only ? nullptr : (float*)n->mask7.p, R, P, P, C5, n->box7_sm.p,
                                 n->mask7_sm.p, sm_box));
  // box-feature branch (test.prototxt:604-652): on the second stream when it is available, otherwise in line
  const bool fork = n->ctx_b && ctx->profiling == 0 && R > 0;
  mnc_ctx* cb = fork ? n->ctx_b : ctx;
  const int si = second ? 1 : 0;
  if (fork) {
    MNC_HIP_TRY(hipEventRecord(n->ev_fork[si], ctx->stream));                                   // feat14 / box7 are complete
    MNC_HIP_TRY(hipStreamWaitEvent(cb->stream, n->ev_fork[si], 0));
  }
  if (!one_pass) NET_TRY(mnc_maxpool2_rhwc_sm(cb, feat14, (float*)n->box7.p, R, P, P, C5, n->box7_sm.p, sm_box));
  const int sm_f6 = sm_format(n, n->fc7, F), sm_f6m = sm_format(n, n->fc7m, F);
  if (n->fuse_small && one_pass && !fork && n->fc6.kind == 0 && n->fc6m.kind == 0 && n->fc7.kind == 0 && n->fc7m.kind == 0 &&
      n->fc6.K == n->fc6m.K) {
    // fp32: the two branches' InnerProducts in pairs (box7 and mask7 both exist since the one-pass pooling): fc6 + fc6_mask, then
    // fc7 + fc7_mask, each ONE launch (mnc_fc_pair; the Python engine pairs the same layers: engine.py, _plan_fusions)
    NET_TRY(mnc_fc_pair(ctx, (const float*)n->box7.p, (const float*)n->fc6.w, n->fc6.b, (float*)n->f6.p, (const float*)n->mask7.p,
                        (const float*)n->fc6m.w, n->fc6m.b, (float*)n->f6m.p, R, F, n->fc6.K, F, 1));
    NET_TRY(mnc_fc_pair(ctx, (const float*)n->f6.p, (const float*)n->fc7.w, n->fc7.b, join + F, (const float*)n->f6m.p,
                        (const float*)n->fc7m.w, n->fc7m.b, join, R, F, F, 2 * F, 1));
  } else if (lowp_pairs) {
    // fp16 / plain bf16 / split bf16 (round 6): the same pairs on the 256-column reduced-precision kernel (mnc_fc_lowp_pair; engine.py
    // pairs the same layers).  Inputs in their stage-major form where the producer wrote it, outputs of fc6 / fc6_mask a second time in
    // fc7's (format 1 = fp16, 2 = split bf16).
    const int mode = n->fc6.kind == 1 ? 0 : n->fc6.kind == 2 ? 1 : 2;
    const bool pre6 = sm_box != 0, pre7 = sm_f6 != 0;
    NET_TRY(mnc_fc_lowp_pair(ctx, mode, pre6 ? nullptr : (const float*)n->box7.p, pre6 ? n->box7_sm.p : nullptr,
                             pre6 ? nullptr : (const float*)n->mask7.p, pre6 ? n->mask7_sm.p : nullptr, R, n->fc6.w, n->fc6m.w, n->fc6.b,
                             n->fc6m.b, (float*)n->f6.p, (float*)n->f6m.p, R, F, n->fc6.K, F, 1, pre7 ? n->f6_sm.p : nullptr,
                             pre7 ? n->f6m_sm.p : nullptr, sm_f6));
    NET_TRY(mnc_fc_lowp_pair(ctx, mode, pre7 ? nullptr : (const float*)n->f6.p, pre7 ? n->f6_sm.p : nullptr,
                             pre7 ? nullptr : (const float*)n->f6m.p, pre7 ? n->f6m_sm.p : nullptr, R, n->fc7.w, n->fc7m.w, n->fc7.b,
                             n->fc7m.b, join + F, join, R, F, F, 2 * F, 1, nullptr, nullptr, 0));
  } else {
  NET_TRY(run_fc_sm(cb, n->fc6, (const float*)n->box7.p, n->box7_sm.p, sm_box, (float*)n->f6.p, R, F, 1, n->f6_sm.p, sm_f6));
  NET_TRY(run_fc_sm(cb, n->fc7, (const float*)n->f6.p, n->f6_sm.p, n->fc6.kind ? sm_f6 : 0, join + F, R, 2 * F, 1));
  if (fork) MNC_HIP_TRY(hipEventRecord(n->ev_join[si], cb->stream));
  if (!one_pass)
    NET_TRY(mnc_mask_pool_sm(ctx, feat14, (const float*)n->m14.p, (float*)n->mask7.p, R, P, P, C5, 1, n->mask7_sm.p, sm_mask));
  NET_TRY(run_fc_sm(ctx, n->fc6m, (const float*)n->mask7.p, n->mask7_sm.p, sm_mask, (float*)n->f6m.p, R, F, 1, n->f6m_sm.p, sm_f6m));
  NET_TRY(run_fc_sm(ctx, n->fc7m, (const float*)n->f6m.p, n->f6m_sm.p, n->fc6m.kind ? sm_f6m : 0, join, R, 2 * F, 1));
  if (fork) MNC_HIP_TRY(hipStreamWaitEvent(ctx->stream, n->ev_join[si], 0));
  }
  float* heads = (float*)n->heads.p + (size_t)row0 * 6 * K;      // kept per stage: mnc_net_blob("head_scores")
  float* scores = (float*)n->scores.p + (size_t)row0 * K;
  n->tail_done = false;
  if (n->fuse_small && R > 0 && K <= 64 && n->fc_heads.kind == 0 && 2.0 * R * 6.0 * K * n->fc_heads.K < 2.0e9) {
    // the sibling classifiers' GEMM leaves its K ranges; ONE kernel sums them (+ bias), takes the seg_cls_score softmax and runs
    // the stage bridge (stage 3) or im_detect's tail (stage 5) on the row -- the bits of the four separate launches below
    ctx->defer_reduce = true;
    int rc = run_fc(n, n->fc_heads, join, heads, R, 6 * K, 0);
    ctx->defer_reduce = false;
    if (rc) return rc;
    NET_TRY(heads_finish_launch(ctx, ctx->deferred_part, ctx->deferred_splits, n->fc_heads.b, heads, 6 * K, R, K, scores, rois,
                                (float)n->OH, (float)n->OW, second ? nullptr : (float*)n->rois_ext.p, (const float*)n->rois.p,
                                (const float*)n->rois_ext.p, n->im_scale, n->H, n->W, second ? (float*)n->boxes.p : nullptr,
                                second ? proposal_count_ptr(ctx) : nullptr, second ? n->prop_count() : nullptr));
    n->tail_done = second;
    return MNC_OK;
  }
  NET_TRY(run_fc(n, n->fc_heads, join, heads, R, 6 * K, 0));
  if (R) NET_TRY(mnc_softmax_rows_ld(ctx, heads + K, 6 * K, scores, R, K));                     // seg_cls_prob
  if (!second)
    NET_TRY(mnc_stage_bridge(ctx, rois, heads + 2 * K, 6 * K, scores, K, R, K, (float)n->OH, (float)n->OW,
                             (float*)n->rois_ext.p));
  return MNC_OK;
}

// heads of both stages on r1 proposals, im_detect's tail, voting; records + counts + the proposal count to the pinned buffer
int run_heads_and_vote(mnc_net* n, int r1) {
  const mnc_net_config& c = n->cfg;
  mnc_ctx* ctx = n->ctx;
  const int S = c.mask_size, K = c.num_classes;
  NET_TRY(run_stage(n, (const float*)n->rois.p, r1, false, 0));
  NET_TRY(run_stage(n, (const float*)n->rois_ext.p, r1, true, r1));
  if (!n->tail_done)             // (otherwise the stage-5 finish kernel wrote the boxes and moved the proposal count)
    NET_TRY(detect_tail_launch(ctx, (const float*)n->rois.p, r1, (const float*)n->rois_ext.p, r1, n->im_scale, n->H, n->W,
                               (float*)n->boxes.p, proposal_count_ptr(ctx), n->prop_count()));
  const int rows = (K - 1) * c.max_per_image;
  NET_TRY(mnc_vote_instances(ctx, (const float*)n->boxes.p, (const float*)n->masks.p, (const float*)n->scores.p, 2 * r1, K, S,
                             c.max_per_image, c.vote_nms_thresh, c.vote_iou_thresh, n->H, n->W, n->records(), rows, n->counts()));
  n->last_r1 = r1; n->last_r2 = r1;
  return MNC_OK;
}

int enqueue_outputs(mnc_net* n) {
  const mnc_net_config& c = n->cfg;
  const size_t rec_bytes = (size_t)c.max_per_image * (6 + c.mask_size * c.mask_size) * 4;     // the first max_per_image rows
  // [counts | proposal count (moved into the block by the tail kernel) | the first max_per_image records]: one copy
  NET_TRY(mnc_d2h_async(n->ctx, n->pin_out, n->outblk.p, 512 + rec_bytes));
  return MNC_OK;
}

int enqueue_image(mnc_net* n) {
  NET_TRY(mnc_h2d_async(n->ctx, n->img.p, n->pin_img, (size_t)n->H * n->W * 3));
  NET_TRY(run_trunk(n));
  NET_TRY(run_heads_and_vote(n, n->cfg.post_nms_topn));
  return enqueue_outputs(n);
}

unsigned long arena_gen(const mnc_net* n) { return n->ctx->arena_gen + (n->ctx_b ? n->ctx_b->arena_gen : 0); }

// The asynchronous part of one image: eager the first time a size is seen (buffers, scratch and function attributes settle),
// captured into a graph the second time, replayed from then on.
int launch_image(mnc_net* n, const unsigned char* bgr_host, int H, int W) {
  MNC_REQUIRE(n && bgr_host && H >= 16 && W >= 16, "mnc_forward_image: bad argument");
  NET_TRY(finalize(n));
  MNC_HIP_TRY(hipSetDevice(n->ctx->device));
  if (n->in_flight) MNC_HIP_TRY(hipStreamSynchronize(n->ctx->stream));   // launch without fetch: the staged image must be consumed
  NET_TRY(set_geometry(n, H, W));
  memcpy(n->pin_img, bgr_host, (size_t)H * W * 3);
  n->in_flight = true;
  hipStream_t s = n->ctx->stream;
  const bool want_graph = n->cfg.use_graph && n->ctx->profiling == 0;
  if (n->gexec && n->graph_gen != arena_gen(n)) {
    // a context-owned arena (split-K / Winograd scratch, proposal state, voting scratch) was re-allocated since the capture --
    // by another image size run eagerly on this net (the scratch need is not monotonic in the image size) or by another user of
    // the context: the graph holds freed addresses.  Drop it; this image runs as direct launches, the next one re-captures.
    (void)hipGraphExecDestroy(n->gexec);
    n->gexec = nullptr; n->graph_h = n->graph_w = -1; n->seen_h = n->seen_w = -1;
  }
  if (want_graph && n->gexec && n->graph_h == H && n->graph_w == W) {
    MNC_HIP_TRY(hipGraphLaunch(n->gexec, s));
    return MNC_OK;
  }
  if (want_graph && n->seen_h == H && n->seen_w == W) {
    if (n->gexec) { (void)hipGraphExecDestroy(n->gexec); n->gexec = nullptr; }
    hipGraph_t g = nullptr;
    MNC_HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    const unsigned long gen0 = arena_gen(n);
    n->ctx->capturing = true;          // an arena that would have to grow now fails its launch instead of synchronising the stream
    if (n->ctx_b) n->ctx_b->capturing = true;
    int rc = enqueue_image(n);
    n->ctx->capturing = false;
    if (n->ctx_b) n->ctx_b->capturing = false;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc == MNC_OK && e == hipSuccess && g && gen0 == arena_gen(n)) {
      e = hipGraphInstantiate(&n->gexec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e == hipSuccess) {
        n->graph_h = H; n->graph_w = W;
        n->graph_gen = gen0;
        MNC_HIP_TRY(hipGraphLaunch(n->gexec, s));
        return MNC_OK;
      }
      n->gexec = nullptr;
      (void)hipGetLastError();
      n->cfg.use_graph = 0;            // instantiation is not available here: stay on direct launches
    } else {
      // the sequence could not be captured as it stands (an arena had to grow -- cannot happen after an eager run of the same size,
      // kept as a guard -- or the runtime refused): discard what was recorded; this image runs directly and the size is captured
      // again on its next image
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      if (e != hipSuccess) n->cfg.use_graph = 0;
    }
  }
  n->seen_h = H; n->seen_w = W;
  return enqueue_image(n);
}

}  // namespace

extern "C" {

int mnc_net_default_config(mnc_net_config* cfg) {
  MNC_REQUIRE(cfg, "mnc_net_default_config: null pointer");
  memset(cfg, 0, sizeof(*cfg));
  const int tc[5] = {64, 128, 256, 512, 512};
  memcpy(cfg->trunk_channels, tc, sizeof(tc));
  cfg->rpn_channels = 512;
  cfg->num_anchors = 9;
  // transform.anchors.generate_anchors(): ratios {0.5, 1, 2} x scales {8, 16, 32} around (0, 0, 15, 15)  (anchors.py:38-49)
  const float a[36] = {-84, -40, 99, 55,  -176, -88, 191, 103,  -360, -184, 375, 199,  -56, -56, 71, 71,  -120, -120, 135, 135,
                       -248, -248, 263, 263,  -36, -80, 51, 95,  -80, -168, 95, 183,  -168, -344, 183, 359};
  memcpy(cfg->anchors, a, sizeof(a));
  cfg->feat_stride = 16;
  cfg->pre_nms_topn = 6000; cfg->post_nms_topn = 300;
  cfg->rpn_nms_thresh = 0.7f; cfg->rpn_min_size = 16.0f;
  cfg->mask_fc = 256; cfg->mask_size = 21;
  cfg->fc_dim = 4096;
  cfg->num_classes = 21;
  cfg->roi_size = 14;
  cfg->spatial_scale = 0.0625f;
  cfg->target_size = 600; cfg->max_size = 1000;
  cfg->pixel_means[0] = 102.9801; cfg->pixel_means[1] = 115.9465; cfg->pixel_means[2] = 122.7717;
  cfg->max_per_image = 100;
  cfg->vote_nms_thresh = 0.3f; cfg->vote_iou_thresh = 0.5f;
  cfg->math = 0;
  cfg->use_graph = 1;
  cfg->winograd = 4;
  cfg->conventions.maskpool_thresh = 0.4f;         // every switch 0: oracle/SPEC.md ...
  cfg->conventions.inherit = 1;                    // ... but the context's conventions stay in force unless the host clears this
  clear_error();
  return MNC_OK;
}

int mnc_net_create(mnc_ctx* ctx, const mnc_net_config* cfg, mnc_net** out) {
  MNC_REQUIRE(ctx && cfg && out, "mnc_net_create: null pointer");
  *out = nullptr;
  for (int i = 0; i < 5; ++i)
    MNC_REQUIRE(cfg->trunk_channels[i] > 0 && cfg->trunk_channels[i] % 8 == 0 && cfg->trunk_channels[i] <= 4096,
                "mnc_net_create: trunk width %d must be a positive multiple of 8", cfg->trunk_channels[i]);
  MNC_REQUIRE(cfg->trunk_channels[0] <= 512, "mnc_net_create: conv1 width %d > 512", cfg->trunk_channels[0]);
  MNC_REQUIRE(cfg->rpn_channels > 0 && cfg->rpn_channels % 8 == 0 && cfg->num_anchors > 0 && cfg->num_anchors <= 16,
              "mnc_net_create: RPN shape");
  MNC_REQUIRE(cfg->roi_size > 0 && cfg->roi_size % 2 == 0 && cfg->mask_size >= 2 && cfg->num_classes >= 2 &&
              cfg->post_nms_topn > 0 && cfg->max_per_image > 0 && cfg->fc_dim > 0 && cfg->mask_fc > 0,
              "mnc_net_create: head shape");
  // the per-class counts travel in a fixed 256-byte header (device `counts` buffer, pinned [counts | proposal count | records])
  MNC_REQUIRE(cfg->num_classes <= 64, "mnc_net_create: num_classes %d > 64 (the result header holds 64 counts)", cfg->num_classes);
  MNC_REQUIRE(cfg->math >= 0 && cfg->math <= 4, "mnc_net_create: math must be 0 (fp32), 1 (bf16x3), 2 (f16), 3 (mixed) or 4 (bf16)");
  MNC_REQUIRE(cfg->target_size > 0 && cfg->max_size >= cfg->target_size, "mnc_net_create: target_size / max_size");
  // The RoI launchers read the conventions from the context.  conventions.inherit (set by mnc_net_default_config) keeps what is
  // in force there; anything else is validated and applied -- an explicit all-zero member included (the all-zero value
  // used to double as "inherit", so a host could not ask for the SPEC on a context an earlier net had changed).
  if (!cfg->conventions.inherit) {
    int rc = mnc_ctx_set_layer_conventions(ctx, &cfg->conventions);
    if (rc) return rc;
  }
  // F(4x4,3x3) addresses its operands through 32-bit buffer offsets (conv_wino4.hip): the largest trunk map must stay below 2 GB
  // (a 64-channel map of 8.4 Mpixel).  Refused here, with the remedy, rather than at the first oversized image.
  MNC_REQUIRE(cfg->math != 0 || cfg->winograd != 4 ||
                  (double)cfg->trunk_channels[0] * cfg->max_size * (double)cfg->max_size * 4.0 < 2147483648.0,
              "mnc_net_create: winograd = 4 with max_size %d: a %d-channel %dx%d map exceeds the F(4x4) kernel's 2 GB operand range; "
              "use winograd = 2", cfg->max_size, cfg->trunk_channels[0], cfg->max_size, cfg->max_size);
  mnc_net* n = new (std::nothrow) mnc_net();
  if (!n) { set_error("mnc_net_create: out of host memory"); return MNC_ERR_NOMEM; }
  n->ctx = ctx;
  n->cfg = *cfg;
  n->cfg.conventions = ctx->conv;          // the conventions in force (the member, or what the host had set on the context)
  n->fc_sm = tune(ctx, T_FC_SM, 1) != 0;
  n->fuse_pools = tune(ctx, T_FUSE_POOLS, 1) != 0;
  n->fuse_small = tune(ctx, T_FUSE_SMALL, 1) != 0;
  if (tune(ctx, T_BRANCH_STREAMS, 0) == 1) {
    if (mnc_ctx_create(&n->ctx_b, ctx->device) != MNC_OK) n->ctx_b = nullptr;     // optional: the net works on one stream
    if (n->ctx_b) (void)mnc_ctx_set_layer_conventions(n->ctx_b, &ctx->conv);
    for (int i = 0; i < 2 && n->ctx_b; ++i) {
      if (hipEventCreateWithFlags(&n->ev_fork[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&n->ev_join[i], hipEventDisableTiming) != hipSuccess) {
        (void)mnc_ctx_destroy(n->ctx_b);
        n->ctx_b = nullptr;
      }
    }
  }
  *out = n;
  clear_error();
  return MNC_OK;
}

int mnc_net_create_shared(mnc_ctx* ctx, mnc_net* parent, mnc_net** out) {
  MNC_REQUIRE(ctx && parent && out, "mnc_net_create_shared: null pointer");
  *out = nullptr;
  MNC_REQUIRE(!parent->owner, "mnc_net_create_shared: `parent` borrows its weights itself; share with the net that owns them");
  MNC_REQUIRE(ctx->device == parent->ctx->device, "mnc_net_create_shared: context on device %d, the weights live on device %d",
              ctx->device, parent->ctx->device);
  MNC_REQUIRE(ctx != parent->ctx, "mnc_net_create_shared: give the sharing net its own context (stream and scratch arenas)");
  NET_TRY(finalize(parent));                       // packs and uploads once; synchronises the parent's stream
  mnc_net* n = nullptr;
  {
    mnc_net_config cfg = parent->cfg;              // (the conventions in force on the parent's context included)
    int rc = mnc_net_create(ctx, &cfg, &n);
    if (rc) return rc;
    rc = mnc_ctx_set_layer_conventions(ctx, &parent->cfg.conventions);
    if (rc) { (void)mnc_net_destroy(n); return rc; }
    n->cfg.conventions = ctx->conv;
  }
  n->owner = parent;
  ++parent->sharers;
  n->w_c3 = parent->w_c3;
  for (int i = 0; i < 14; ++i) {
    n->w_conv[i] = parent->w_conv[i];
    n->b_conv[i] = parent->b_conv[i];
    n->conv_fast[i] = parent->conv_fast[i];
  }
  n->packed_trunk = parent->packed_trunk;
  n->w_rpn = parent->w_rpn; n->b_rpn = parent->b_rpn;
  n->fc_maskest = parent->fc_maskest; n->fc_maskpred = parent->fc_maskpred;
  n->fc6 = parent->fc6; n->fc7 = parent->fc7; n->fc6m = parent->fc6m; n->fc7m = parent->fc7m; n->fc_heads = parent->fc_heads;
  n->finalized = true;
  *out = n;
  clear_error();
  return MNC_OK;
}

int mnc_net_set_param(mnc_net* net, const char* layer, int index, const float* data_host, size_t count) {
  MNC_REQUIRE(net && layer && data_host && index >= 0 && index <= 1 && count > 0, "mnc_net_set_param: bad argument");
  MNC_REQUIRE(!net->owner, "mnc_net_set_param: this net shares another net's weights (mnc_net_create_shared)");
  MNC_REQUIRE(!net->finalized, "mnc_net_set_param: the weights of this net are already packed (set them before the first image)");
  try {
    HostBlob& b = net->params[std::string(layer) + "/" + std::to_string(index)];
    b.v.assign(data_host, data_host + count);
  } catch (const std::exception&) {
    set_error("mnc_net_set_param: out of host memory for %s/%d (%zu values)", layer, index, count);
    return MNC_ERR_NOMEM;
  }
  clear_error();
  return MNC_OK;
}

int mnc_net_load_file(mnc_net* net, const char* path) {
  MNC_REQUIRE(net && path, "mnc_net_load_file: null pointer");
  FILE* f = fopen(path, "rb");
  MNC_REQUIRE(f, "mnc_net_load_file: cannot open %s", path);
  // the file is untrusted input: every count is checked against what is left of the file before anything is allocated, and
  // nothing may throw across the C boundary
  long file_bytes = 0;
  if (fseek(f, 0, SEEK_END) == 0) file_bytes = ftell(f);
  rewind(f);
  char magic[8];
  unsigned n = 0;
  bool ok = file_bytes >= 12 && fread(magic, 1, 8, f) == 8 && memcmp(magic, "MNCW0001", 8) == 0 && fread(&n, 4, 1, f) == 1;
  int rc = MNC_OK;
  const char* why = "is not a complete MNCW0001 file";
  try {
    std::vector<float> v;
    for (unsigned i = 0; ok && i < n; ++i) {
      unsigned short len = 0;
      unsigned char idx = 0, nd = 0;
      char name[256];
      unsigned dims[8];
      ok = fread(&len, 2, 1, f) == 1 && len < sizeof(name) && fread(name, 1, len, f) == len && fread(&idx, 1, 1, f) == 1 &&
           fread(&nd, 1, 1, f) == 1 && nd <= 8 && fread(dims, 4, nd, f) == nd;
      if (!ok) break;
      name[len] = 0;
      const bool skip = idx > 1;     // a third blob (BatchNorm's moving-average factor in containers save_flat wrote): not a parameter of this graph
      const size_t left = (size_t)(file_bytes - ftell(f)) / 4;
      size_t count = 1;
      for (int d = 0; d < nd && ok; ++d) {
        if (dims[d] == 0 || count > left / dims[d]) ok = false;      // empty blob, or more values than the file has left
        else count *= dims[d];
      }
      if (!ok) { why = "declares a blob larger than the file"; break; }
      v.resize(count);
      ok = fread(v.data(), 4, count, f) == count;
      if (!ok) break;
      if (!skip) rc = mnc_net_set_param(net, name, idx, v.data(), count);
      if (rc) break;
    }
  } catch (const std::exception&) {
    fclose(f);
    set_error("mnc_net_load_file: out of host memory while reading %s", path);
    return MNC_ERR_NOMEM;
  }
  fclose(f);
  if (rc) return rc;
  MNC_REQUIRE(ok, "mnc_net_load_file: %s %s", path, why);
  clear_error();
  return MNC_OK;
}

int mnc_forward_image_async(mnc_net* net, const unsigned char* bgr_host, int H, int W, float** d_records, int** d_counts) {
  int rc = launch_image(net, bgr_host, H, W);
  if (rc) return rc;
  if (d_records) *d_records = net->records();
  if (d_counts) *d_counts = net->counts();
  clear_error();
  return MNC_OK;
}

int mnc_net_fetch(mnc_net* net, float* records_host, int record_cap, int* counts_host) {
  MNC_REQUIRE(net && records_host && counts_host && record_cap >= 0, "mnc_net_fetch: null pointer");
  MNC_REQUIRE(net->H > 0, "mnc_net_fetch: no image has been launched on this net");
  const mnc_net_config& c = net->cfg;
  const int D = 6 + c.mask_size * c.mask_size;
  MNC_REQUIRE(record_cap <= (c.num_classes - 1) * c.max_per_image, "mnc_net_fetch: record_cap %d > %d", record_cap,
              (c.num_classes - 1) * c.max_per_image);
  MNC_HIP_TRY(hipSetDevice(net->ctx->device));
  MNC_HIP_TRY(hipStreamSynchronize(net->ctx->stream));
  net->in_flight = false;
  const int* head = (const int*)net->pin_out;
  const int n_prop = *(const int*)((const char*)net->pin_out + 256);
  if (n_prop < c.post_nms_topn) {
    // fewer proposals survived the RPN's NMS than post_nms_topn: the speculative rows were zero boxes; the reference runs the
    // heads on exactly n_prop rois -- so do that (direct launches; rare, and never with a trained RPN on a natural image)
    NET_TRY(run_heads_and_vote(net, n_prop));
    NET_TRY(enqueue_outputs(net));
    MNC_HIP_TRY(hipStreamSynchronize(net->ctx->stream));
  }
  const int R = head[0];
  for (int k = 0; k < c.num_classes; ++k) counts_host[k] = head[k];
  const int first = c.max_per_image < record_cap ? c.max_per_image : record_cap;
  const int have = R < first ? R : first;
  memcpy(records_host, (const char*)net->pin_out + 512, (size_t)have * D * 4);
  if (have < record_cap) memset(records_host + (size_t)have * D, 0, (size_t)(record_cap - have) * D * 4);
  if (R > first && record_cap > first) {          // scores tied at the voting threshold: rows beyond max_per_image
    const int more = (R < record_cap ? R : record_cap) - first;
    NET_TRY(mnc_d2h(net->ctx, records_host + (size_t)first * D, (const float*)net->records() + (size_t)first * D,
                    (size_t)more * D * 4));
  }
  clear_error();
  return MNC_OK;
}

int mnc_forward_image(mnc_net* net, const unsigned char* bgr_host, int H, int W, float* records_host, int record_cap,
                      int* counts_host) {
  MNC_REQUIRE(net && records_host && counts_host && record_cap >= 0, "mnc_forward_image: null pointer");
  int rc = launch_image(net, bgr_host, H, W);
  if (rc) return rc;
  return mnc_net_fetch(net, records_host, record_cap, counts_host);
}

int mnc_net_blob(mnc_net* net, const char* name, void** d_ptr, int* dims, int* ndim) {
  MNC_REQUIRE(net && name && d_ptr && dims && ndim, "mnc_net_blob: null pointer");
  const mnc_net_config& c = net->cfg;
  const std::string s(name);
  const int A = c.num_anchors, R1 = net->last_r1, R2 = net->last_r2;
  auto set = [&](void* p, int n, int a, int b, int cc, int d) {
    *d_ptr = p; *ndim = n; dims[0] = a; dims[1] = b; dims[2] = cc; dims[3] = d;
    return MNC_OK;
  };
  if (s == "data") return set(net->data.p, 4, 1, 3, net->OH, net->OW);
  if (s == "conv5_3") return set(net->act[12].p, 4, 1, c.trunk_channels[4], net->fh, net->fw);
  if (s == "rpn_cls_prob_reshape") return set(net->rpn_prob.p, 4, 1, 2 * A, net->fh, net->fw);
  if (s == "rpn_bbox_pred") return set(net->rpn_bbox_p, 4, 1, 4 * A, net->fh, net->fw);
  if (s == "rois") return set(net->rois.p, 2, R1, 5, 0, 0);
  if (s == "rois_ext") return set(net->rois_ext.p, 2, R2, 5, 0, 0);
  if (s == "mask_proposal") return set(net->masks.p, 4, R1 + R2, 1, c.mask_size, c.mask_size);
  if (s == "seg_cls_prob") return set(net->scores.p, 2, R1 + R2, c.num_classes, 0, 0);
  if (s == "boxes") return set(net->boxes.p, 2, R1 + R2, 4, 0, 0);
  // rows of both stages, columns [cls_score (K) | seg_cls_score (K) | bbox_pred (4K)]: the three sibling InnerProducts are one GEMM
  if (s == "head_scores") return set(net->heads.p, 2, R1 + R2, 6 * c.num_classes, 0, 0);
  if (s == "records") return set(net->records(), 2, (c.num_classes - 1) * c.max_per_image, 6 + c.mask_size * c.mask_size, 0, 0);
  set_error("mnc_net_blob: unknown blob %s", name);
  return MNC_ERR_INVALID;
}

int mnc_net_destroy(mnc_net* net) {
  if (!net) return MNC_OK;
  if (net->sharers > 0) {
    set_error("mnc_net_destroy: %d net(s) created with mnc_net_create_shared still use this net's weights; destroy them first",
              net->sharers.load());
    return MNC_ERR_STATE;
  }
  mnc_ctx* ctx = net->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (net->gexec) (void)hipGraphExecDestroy(net->gexec);
  DevBuf* bufs[] = {&net->img, &net->taps, &net->data, &net->rpn_out, &net->rpn_score, &net->rpn_prob, &net->rois,
                    &net->rois_ext, &net->feat14, &net->h_mask, &net->m14, &net->box7, &net->mask7, &net->f6, &net->f6m, &net->join,
                    &net->feat14_sm, &net->box7_sm, &net->mask7_sm, &net->f6_sm, &net->f6m_sm,
                    &net->heads, &net->boxes, &net->masks, &net->scores, &net->outblk, &net->hwc5, &net->act12_pk};
  for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
  for (auto& b : net->act) if (b.p) (void)hipFree(b.p);
  for (auto& b : net->pooled) if (b.p) (void)hipFree(b.p);
  if (net->owner) {
    --net->owner->sharers;                       // borrowed weights: nothing to free
  } else {
    void* singles[] = {net->w_c3, net->w_rpn, net->b_rpn};
    for (void* p : singles) if (p) (void)hipFree(p);
    for (int i = 0; i < 14; ++i) {
      if (net->w_conv[i]) (void)hipFree(net->w_conv[i]);
      if (net->b_conv[i]) (void)hipFree(net->b_conv[i]);
    }
    mnc_net::Fc* fcs[] = {&net->fc_maskest, &net->fc_maskpred, &net->fc6, &net->fc7, &net->fc6m, &net->fc7m, &net->fc_heads};
    for (mnc_net::Fc* f : fcs) {
      if (f->w) (void)hipFree(f->w);
      if (f->b) (void)hipFree(f->b);
    }
  }
  if (net->pin_img) (void)hipHostFree(net->pin_img);
  if (net->pin_out) (void)hipHostFree(net->pin_out);
  for (int i = 0; i < 2; ++i) {
    if (net->ev_fork[i]) (void)hipEventDestroy(net->ev_fork[i]);
    if (net->ev_join[i]) (void)hipEventDestroy(net->ev_join[i]);
  }
  if (net->ctx_b) (void)mnc_ctx_destroy(net->ctx_b);
  delete net;
  clear_error();
  return MNC_OK;
}

}  // extern "C"
