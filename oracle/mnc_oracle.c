/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the native pieces of MNC's inference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  Nothing under mnc_amd/ may import, link or call it.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).  Compile with
 * -ffp-contract=off (oracle/Makefile does): the float expressions below are evaluated exactly as written,
 * one IEEE-754 binary32 operation per C operator, which is also how the HIP kernels are compiled.
 *
 * Pinning status:
 *   orc_nms / orc_mv / orc_bbox_overlaps : pinned against the reference's own code -- oracle/_ref/libmnc_ref.so is
 *       lib/nms/nms_kernel.cu + lib/nms/mv_kernel.cu compiled for the CPU (oracle/build_ref.py); see
 *       tests/test_oracle_vs_ref.py, and the committed fixtures in tests/golden/.
 *   orc_roi_warp / orc_mask_resize / orc_mask_pool / orc_maxpool2 / orc_roi_pool : PARITY UNPINNED.  Their arithmetic lives in the
 *       un-vendored `caffe-mnc` submodule (.gitmodules:1-3, pinned SHA unknown, directory empty).  They follow
 *       oracle/SPEC.md; every convention that had to be chosen is marked SPEC-CHOICE.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

static inline float fmax_(float a, float b) { return a > b ? a : b; }
static inline float fmin_(float a, float b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------------------
 * NMS -- lib/nms/nms_kernel.cu
 * ---------------------------------------------------------------------------------------------- */

/* devIoU, lib/nms/nms_kernel.cu:24-32.  +1 "inclusive pixel" widths, each side of the intersection
 * clamped at 0, interS / (Sa + Sb - interS) in that order. */
static float orc_iou(const float* a, const float* b) {
  float left = fmax_(a[0], b[0]), right = fmin_(a[2], b[2]);
  float top = fmax_(a[1], b[1]), bottom = fmin_(a[3], b[3]);
  float width = fmax_(right - left + 1, 0.f), height = fmax_(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

/* The 64x64-tiled suppression bitmask, lib/nms/nms_kernel.cu:34-78: bit j of word (i, c) is set iff
 * IoU(box i, box 64c+j) > thresh (strict); inside diagonal tiles only j > i (:66-69).
 * mask has n * ceil(n/64) words.  Exposed so tests can compare the device bitmask word for word. */
ORC_EXPORT void orc_nms_mask(const float* boxes, int n, int dim, float thresh, unsigned long long* mask) {
  const int cb = (n + 63) / 64;
  for (int i = 0; i < n; ++i) {
    const int rb = i / 64;
    for (int c = 0; c < cb; ++c) {
      unsigned long long t = 0;
      const int csize = (n - c * 64) < 64 ? (n - c * 64) : 64;
      int start = (rb == c) ? (i % 64) + 1 : 0;
      for (int j = start; j < csize; ++j)
        if (orc_iou(boxes + (long)i * dim, boxes + (long)(c * 64 + j) * dim) > thresh) t |= 1ULL << j;
      mask[(long)i * cb + c] = t;
    }
  }
}

/* _nms, lib/nms/nms_kernel.cu:91-144: boxes are ALREADY sorted by descending score (gpu_nms.pyx:26-29);
 * greedy scan (:124-140): keep i iff its bit in remv is clear, then remv |= row_i from word i/64 on.
 * keep_out has capacity n; indices are positions in the sorted array. */
ORC_EXPORT void orc_nms(int* keep_out, int* num_out, const float* boxes, int n, int dim, float thresh) {
  const int cb = (n + 63) / 64;
  unsigned long long* mask = (unsigned long long*)malloc((size_t)(n > 0 ? n : 1) * (cb > 0 ? cb : 1) * 8);
  unsigned long long* remv = (unsigned long long*)calloc(cb > 0 ? cb : 1, 8);
  orc_nms_mask(boxes, n, dim, thresh, mask);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    const int nb = i / 64, ib = i % 64;
    if (!(remv[nb] & (1ULL << ib))) {
      keep_out[nk++] = i;
      const unsigned long long* p = mask + (long)i * cb;
      for (int j = nb; j < cb; ++j) remv[j] |= p[j];
    }
  }
  *num_out = nk;
  free(mask);
  free(remv);
}

/* ------------------------------------------------------------------------------------------------
 * bbox_overlaps -- lib/utils/bbox.pyx:15-55 (float64, +1 widths, 0 when the boxes do not intersect)
 * ---------------------------------------------------------------------------------------------- */
ORC_EXPORT void orc_bbox_overlaps(const double* boxes, int N, const double* query, int K, double* out) {
  for (long i = 0; i < (long)N * K; ++i) out[i] = 0.0;
  for (int k = 0; k < K; ++k) {
    const double* q = query + 4 * k;
    double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    for (int n = 0; n < N; ++n) {
      const double* b = boxes + 4 * n;
      double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
      if (iw > 0) {
        double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) {
          double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
          out[(long)n * K + k] = iw * ih / ua;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Mask voting -- lib/nms/mv_kernel.cu
 * ---------------------------------------------------------------------------------------------- */

/* mask_render for ONE pixel of ONE input mask, lib/nms/mv_kernel.cu:36-91.
 * Inside test on the un-rounded float box (:52); ratio = S / (x2 - x1 + 1.0) (:56-59, the +1.0 is a double
 * constant, the sum is rounded to float on assignment); top-left aligned bilinear, no half-pixel offset;
 * sx == S-1 or sy == S-1 -> nearest (the three branches :66-73 all read mask[sy*S+sx]). */
static float orc_render_px(const float* box, const float* mask, int S, int h, int w) {
  const float x1 = box[0], y1 = box[1], x2 = box[2], y2 = box[3];
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

/* _mv, lib/nms/mv_kernel.cu:242-348, result by result instead of buffer by buffer (same arithmetic, O(H*W)
 * scratch instead of the reference's N*H*W render buffer):
 *   aggregate[h,w] = sum over the result's candidates, IN CANDIDATE ORDER, of render(cand)*weight (:93-112;
 *                    candidate_start[r] is the END offset of result r, start is candidate_start[r-1], 0 for r==0)
 *   col/row "any(value > 0.4f)" (:114-142, strict, float constant :13)
 *   first/last true index, default W/2 resp. H/2 (integer division) when empty (:144-190)
 *   resample the tight box to S x S, top-left aligned, edge cases tested against the IMAGE border (:193-240)
 * out_box rows are int32 [x1, y1, x2, y2] (:324-329). */
ORC_EXPORT void orc_mv(const float* all_boxes, const float* all_masks, int all_boxes_num,
                       const int* cand_inds, const int* cand_start, const float* cand_weights, int cand_num,
                       int H, int W, int box_dim, int S, int result_num, float* out_mask, int* out_box) {
  (void)all_boxes_num; (void)cand_num;
  const float BIN = 0.4f;
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < result_num; ++r) {
    float* agg = (float*)malloc((size_t)H * W * sizeof(float));
    const int c0 = r == 0 ? 0 : cand_start[r - 1], c1 = cand_start[r];
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        float val = 0.0f;
        for (int i = c0; i < c1; ++i) {
          const int m = cand_inds[i];
          val += orc_render_px(all_boxes + (long)m * box_dim, all_masks + (long)m * S * S, S, h, w) * cand_weights[i];
        }
        agg[(long)h * W + w] = val;
      }
    int bx1 = W / 2, bx2 = W / 2, by1 = H / 2, by2 = H / 2, found = 0;
    for (int w = 0; w < W && !found; ++w)
      for (int h = 0; h < H; ++h) if (agg[(long)h * W + w] > BIN) { bx1 = w; found = 1; break; }
    found = 0;
    for (int w = W - 1; w >= 0 && !found; --w)
      for (int h = 0; h < H; ++h) if (agg[(long)h * W + w] > BIN) { bx2 = w; found = 1; break; }
    found = 0;
    for (int h = 0; h < H && !found; ++h)
      for (int w = 0; w < W; ++w) if (agg[(long)h * W + w] > BIN) { by1 = h; found = 1; break; }
    found = 0;
    for (int h = H - 1; h >= 0 && !found; --h)
      for (int w = 0; w < W; ++w) if (agg[(long)h * W + w] > BIN) { by2 = h; found = 1; break; }
    out_box[r * 4 + 0] = bx1; out_box[r * 4 + 1] = by1; out_box[r * 4 + 2] = bx2; out_box[r * 4 + 3] = by2;

    const float bw = (float)((double)(bx2 - bx1) + 1.0), bh = (float)((double)(by2 - by1) + 1.0);
    const float rw = bw / (float)S, rh = bh / (float)S;
    for (int h = 0; h < S; ++h)
      for (int w = 0; w < S; ++w) {
        const float ix = bx1 + (float)w * rw, iy = by1 + (float)h * rh;
        const int sx = (int)floorf(ix), sy = (int)floorf(iy);
        float v;
        if (sx == W - 1 && sy == H - 1) v = agg[(long)W * H - 1];
        else if (sx == W - 1 || sy == H - 1) v = agg[(long)sy * W + sx];
        else {
          const long tl = (long)sy * W + sx, tr = tl + 1, bl = tl + W, br = bl + 1;
          const float fx = ix - sx, fy = iy - sy;
          const float wtl = (1 - fx) * (1 - fy), wtr = fx * (1 - fy), wbl = (1 - fx) * fy, wbr = fx * fy;
          v = wtl * agg[tl] + wtr * agg[tr] + wbl * agg[bl] + wbr * agg[br];
        }
        out_mask[((long)r * S + h) * S + w] = v;
      }
    free(agg);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Caffe-side MNC layers (source in the missing caffe-mnc submodule) -- oracle/SPEC.md.  PARITY UNPINNED.
 * All tensors NCHW float32, batch index of every RoI is 0 (proposal_layer.py:159-160).
 * ---------------------------------------------------------------------------------------------- */

/* ROIWarping (models/VGG16/mnc_5stage/test.prototxt:479-492 and :809-820).  SPEC.md section 1.
 * SPEC-CHOICE: un-rounded RoI edges x*spatial_scale; roi_w = max(x2s - x1s + 1, 1); bin = roi_w / pooled_w;
 * sample position  x1s + pw*bin  (the MNC paper's  x0 + (u'/W') * w_i ,  arXiv:1512.04412 eq. 5-8, the same
 * top-left alignment as the author's own mv_kernel.cu:36-91);  bilinear kernel kappa(d) = max(0, 1-|d|) over
 * the integer neighbours; taps outside the feature map contribute 0.
 *
 * Every SPEC-CHOICE is a run-time argument of the _ex form (the same switches as `mnc_layer_conventions` of include/mnc_hip.h,
 * evaluated in the same operation order as mnc_amd/csrc/roi.hip), all zero = the SPEC:
 *   sample       0 top-left of the bin: x1s + g*bin          1 bin centre: x1s + (g + 0.5)*bin
 *                2 bin centre in pixel-centre coordinates: x1s + (g + 0.5)*bin - 0.5
 *   round_edges  0 scaled edges as they are                   1 floor(x*scale + 0.5) (ROIPooling's rounding)
 *   no_plus_one  0 roi_w = max(x2s - x1s + 1, 1)              1 roi_w = max(x2s - x1s, 1)
 *   oob          0 taps outside the map contribute 0          1 tap coordinates clamped to the border (replicate) */
ORC_EXPORT void orc_roi_warp_ex(const float* feat, int C, int H, int W, const float* rois, int R,
                                int PH, int PW, float scale, int sample, int round_edges, int no_plus_one, int oob, float* out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + 5 * r;
    float x1s = roi[1] * scale, y1s = roi[2] * scale, x2s = roi[3] * scale, y2s = roi[4] * scale;
    if (round_edges) { x1s = floorf(x1s + 0.5f); y1s = floorf(y1s + 0.5f); x2s = floorf(x2s + 0.5f); y2s = floorf(y2s + 0.5f); }
    const float extra = no_plus_one ? 0.0f : 1.0f;
    const float rw = fmax_(x2s - x1s + extra, 1.0f), rh = fmax_(y2s - y1s + extra, 1.0f);
    const float bw = rw / (float)PW, bh = rh / (float)PH;
    for (int ph = 0; ph < PH; ++ph) {
      const float sy = sample == 0 ? y1s + (float)ph * bh : sample == 1 ? y1s + ((float)ph + 0.5f) * bh
                                                                        : y1s + ((float)ph + 0.5f) * bh - 0.5f;
      const int y0 = (int)floorf(sy);
      const float ay = sy - (float)y0;
      for (int pw = 0; pw < PW; ++pw) {
        const float sx = sample == 0 ? x1s + (float)pw * bw : sample == 1 ? x1s + ((float)pw + 0.5f) * bw
                                                                          : x1s + ((float)pw + 0.5f) * bw - 0.5f;
        const int x0 = (int)floorf(sx);
        const float ax = sx - (float)x0;
        const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay);
        const float w10 = (1.0f - ax) * ay, w11 = ax * ay;
        int vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
        int vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
        int ya = y0, yb = y0 + 1, xa = x0, xb = x0 + 1;
        if (oob) {
          ya = ya < 0 ? 0 : (ya > H - 1 ? H - 1 : ya); yb = yb < 0 ? 0 : (yb > H - 1 ? H - 1 : yb);
          xa = xa < 0 ? 0 : (xa > W - 1 ? W - 1 : xa); xb = xb < 0 ? 0 : (xb > W - 1 ? W - 1 : xb);
          vy0 = vy1 = vx0 = vx1 = 1;
        }
        for (int c = 0; c < C; ++c) {
          const float* f = feat + (long)c * H * W;
          const float f00 = (vy0 && vx0) ? f[(long)ya * W + xa] : 0.0f;
          const float f01 = (vy0 && vx1) ? f[(long)ya * W + xb] : 0.0f;
          const float f10 = (vy1 && vx0) ? f[(long)yb * W + xa] : 0.0f;
          const float f11 = (vy1 && vx1) ? f[(long)yb * W + xb] : 0.0f;
          out[(((long)r * C + c) * PH + ph) * PW + pw] = w00 * f00 + w01 * f01 + w10 * f10 + w11 * f11;
        }
      }
    }
  }
}

ORC_EXPORT void orc_roi_warp(const float* feat, int C, int H, int W, const float* rois, int R,
                             int PH, int PW, float scale, float* out) {
  orc_roi_warp_ex(feat, C, H, W, rois, R, PH, PW, scale, 0, 0, 0, 0, out);
}

/* MaskResize (test.prototxt:558-567, 885-894).  SPEC.md section 2.
 * SPEC-CHOICE (mode 0): the author's own resampling convention from mv_kernel.cu:193-240 -- ratio = in/out, source
 * position = dst*ratio (top-left aligned), floor + bilinear, nearest on the last source row/column.
 * mode 1: half-pixel centres, src = (dst + 0.5)*ratio - 0.5 (cv2.resize / align_corners=False); mode 2: align_corners,
 * src = dst*(in-1)/(out-1).  Modes 1 and 2 clamp src to [0, in-1], take lo = floor(src), hi = min(lo+1, in-1) and blend with the
 * same four products in the same order. */
ORC_EXPORT void orc_mask_resize_ex(const float* in, int R, int IH, int IW, int OH, int OW, int mode, float* out) {
  const float rh = (float)IH / (float)OH, rw = (float)IW / (float)OW;
  const float ah = OH > 1 ? (float)(IH - 1) / (float)(OH - 1) : 0.0f, aw = OW > 1 ? (float)(IW - 1) / (float)(OW - 1) : 0.0f;
  for (int r = 0; r < R; ++r) {
    const float* m = in + (long)r * IH * IW;
    for (int h = 0; h < OH; ++h)
      for (int w = 0; w < OW; ++w) {
        float v;
        if (mode == 0) {
          const float ix = (float)w * rw, iy = (float)h * rh;
          const int sx = (int)floorf(ix), sy = (int)floorf(iy);
          if (sx == IW - 1 || sy == IH - 1) v = m[sy * IW + sx];
          else {
            const float fx = ix - (float)sx, fy = iy - (float)sy;
            v = (1.0f - fx) * (1.0f - fy) * m[sy * IW + sx] + fx * (1.0f - fy) * m[sy * IW + sx + 1] +
                (1.0f - fx) * fy * m[(sy + 1) * IW + sx] + fx * fy * m[(sy + 1) * IW + sx + 1];
          }
        } else {
          float ix = mode == 1 ? ((float)w + 0.5f) * rw - 0.5f : (float)w * aw;
          float iy = mode == 1 ? ((float)h + 0.5f) * rh - 0.5f : (float)h * ah;
          ix = fmin_(fmax_(ix, 0.0f), (float)(IW - 1));
          iy = fmin_(fmax_(iy, 0.0f), (float)(IH - 1));
          const int sx = (int)floorf(ix), sy = (int)floorf(iy);
          const int tx = sx + 1 < IW ? sx + 1 : IW - 1, ty = sy + 1 < IH ? sy + 1 : IH - 1;
          const float fx = ix - (float)sx, fy = iy - (float)sy;
          v = (1.0f - fx) * (1.0f - fy) * m[sy * IW + sx] + fx * (1.0f - fy) * m[sy * IW + tx] +
              (1.0f - fx) * fy * m[ty * IW + sx] + fx * fy * m[ty * IW + tx];
        }
        out[((long)r * OH + h) * OW + w] = v;
      }
  }
}

ORC_EXPORT void orc_mask_resize(const float* in, int R, int IH, int IW, int OH, int OW, float* out) {
  orc_mask_resize_ex(in, R, IH, IW, OH, OW, 0, out);
}

/* MaskPooling (test.prototxt:631-637, 958-964).  SPEC.md section 3.
 * SPEC-CHOICE (binary 0): element-wise product of the per-RoI feature with the continuous mask, broadcast over channels.
 * binary 1: the mask is binarised first, m >= thresh ? 1 : 0 (what the CFM tester feeds, TesterWrapper.py:398). */
ORC_EXPORT void orc_mask_pool_ex(const float* feat, const float* mask, int R, int C, int H, int W, int binary, float thresh,
                                 float* out) {
#pragma omp parallel for
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < H * W; ++i) {
        float mk = mask[(long)r * H * W + i];
        if (binary) mk = mk >= thresh ? 1.0f : 0.0f;
        out[((long)r * C + c) * H * W + i] = feat[((long)r * C + c) * H * W + i] * mk;
      }
}

ORC_EXPORT void orc_mask_pool(const float* feat, const float* mask, int R, int C, int H, int W, float* out) {
  orc_mask_pool_ex(feat, mask, R, C, H, W, 0, 0.0f, out);
}

/* ROIPooling forward (models/VGG16/cfm/test.prototxt:397-407, 446-456).  The layer source sits in the absent caffe-mnc
 * submodule; it is the Fast R-CNN layer (published algorithm, SPEC.md section 4): rounded RoI corners, float32 bin sizes,
 * floor/ceil bin edges clipped to the map, `v > max` from -FLT_MAX, empty bin -> 0.  feat [N][C][H][W], rois [R][5]. */
ORC_EXPORT void orc_roi_pool(const float* feat, int N, int C, int H, int W, const float* rois, int R, int PH, int PW,
                             float scale, float* out) {
#pragma omp parallel for
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (long)r * 5;
    int b = (int)roi[0];
    if (b < 0) b = 0;
    if (b >= N) b = N - 1;
    const int x1 = (int)roundf(roi[1] * scale), y1 = (int)roundf(roi[2] * scale);
    const int x2 = (int)roundf(roi[3] * scale), y2 = (int)roundf(roi[4] * scale);
    const int rw = x2 - x1 + 1 > 1 ? x2 - x1 + 1 : 1, rh = y2 - y1 + 1 > 1 ? y2 - y1 + 1 : 1;
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float* plane = feat + ((long)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
          int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
          hs += y1; he += y1; ws += x1; we += x1;
          hs = hs < 0 ? 0 : (hs > H ? H : hs);
          he = he < 0 ? 0 : (he > H ? H : he);
          ws = ws < 0 ? 0 : (ws > W ? W : ws);
          we = we < 0 ? 0 : (we > W ? W : we);
          float m = (he <= hs || we <= ws) ? 0.f : -3.402823466e38f;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w)
              if (plane[(long)h * W + w] > m) m = plane[(long)h * W + w];
          out[(((long)r * C + c) * PH + ph) * PW + pw] = m;
        }
    }
  }
}

/* Caffe Pooling MAX 2x2 stride 2 pad 0 (BVLC pooling_layer.cpp semantics): output = ceil((n-2)/2)+1, windows
 * clipped to the input (so 75 -> 38, 125 -> 63).  planes = N*C. */
ORC_EXPORT void orc_maxpool2(const float* in, long planes, int H, int W, float* out) {
  const int OH = (H - 2 + 1) / 2 + 1, OW = (W - 2 + 1) / 2 + 1;
#pragma omp parallel for
  for (long p = 0; p < planes; ++p) {
    const float* src = in + p * H * W;
    float* dst = out + p * OH * OW;
    for (int oh = 0; oh < OH; ++oh)
      for (int ow = 0; ow < OW; ++ow) {
        const int h0 = oh * 2, w0 = ow * 2;
        const int h1 = h0 + 2 < H ? h0 + 2 : H, w1 = w0 + 2 < W ? w0 + 2 : W;
        float m = -3.402823466e38f;
        for (int h = h0; h < h1; ++h)
          for (int w = w0; w < w1; ++w) m = fmax_(m, src[(long)h * W + w]);
        dst[(long)oh * OW + ow] = m;
      }
  }
}
