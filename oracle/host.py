"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the host-side (Python) logic on MNC's inference path.

Every function cites the reference lines it follows (paths relative to /root/reference).  The restatement
is checked against the reference's own Python (imported with py2->py3 patches) by
tests/golden/make_golden.py, whose outputs are the committed fixtures in tests/golden/*.npz.

Nothing under mnc_amd/ may import this module.

Numeric note ("numpy-2 promotion"): the reference ran on python-2 / numpy-1.x, where
`float32_array / zero_d_float64_array` stays float32 (value-based casting).  Under numpy >= 2 the same
expression promotes to float64.  The restatement pins the ORIGINAL behaviour by dividing by np.float32(scale).
"""
import numpy as np

from . import native

PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])  # lib/mnc_config.py:20 (BGR)
TEST_SCALE = 600            # lib/mnc_config.py:112  TEST.SCALES
MAX_SIZE = 1000             # lib/mnc_config.py:34   TRAIN.MAX_SIZE (what demo.py:59 actually passes)
RPN_PRE_NMS_TOP_N = 6000    # lib/mnc_config.py:125
RPN_POST_NMS_TOP_N = 300    # lib/mnc_config.py:127
RPN_NMS_THRESH = 0.7        # lib/mnc_config.py:123
RPN_MIN_SIZE = 16           # lib/mnc_config.py:129
MASK_MERGE_IOU_THRESH = 0.5  # lib/mnc_config.py:133
MASK_MERGE_NMS_THRESH = 0.3  # lib/mnc_config.py:134
MASK_SIZE = 21              # lib/mnc_config.py:28


# ---- lib/transform/anchors.py --------------------------------------------------------------------
def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, xc, yc):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """lib/transform/anchors.py:38-102.  Result is the 0-based table in SURVEY.md section 4, NOT the
    1-based MATLAB comment at anchors.py:10-35."""
    ratios = np.asarray(ratios, dtype=np.float64)
    base = np.array([1, 1, base_size, base_size]) - 1
    w, h, xc, yc = _whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mkanchors(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, xc, yc = _whctrs(ratio_anchors[i])
        out.append(_mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out)


# ---- lib/transform/bbox_transform.py -------------------------------------------------------------
def bbox_transform_inv(boxes, deltas):
    """lib/transform/bbox_transform.py:64-99."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx, dy, dw, dh = deltas[:, 0::4], deltas[:, 1::4], deltas[:, 2::4], deltas[:, 3::4]
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = np.exp(dw) * widths[:, None]
    ph = np.exp(dh) * heights[:, None]
    out = np.zeros(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape):
    """lib/transform/bbox_transform.py:102-120 (returns clipped boxes and the indices already inside)."""
    x1, y1, x2, y2 = boxes[:, 0::4], boxes[:, 1::4], boxes[:, 2::4], boxes[:, 3::4]
    keep = np.where((x1 >= 0) & (x2 <= im_shape[1] - 1) & (y1 >= 0) & (y2 <= im_shape[0] - 1))[0]
    out = np.zeros(boxes.shape, dtype=boxes.dtype)
    out[:, 0::4] = np.maximum(np.minimum(x1, im_shape[1] - 1), 0)
    out[:, 1::4] = np.maximum(np.minimum(y1, im_shape[0] - 1), 0)
    out[:, 2::4] = np.maximum(np.minimum(x2, im_shape[1] - 1), 0)
    out[:, 3::4] = np.maximum(np.minimum(y2, im_shape[0] - 1), 0)
    return out, keep


def filter_small_boxes(boxes, min_size):
    """lib/transform/bbox_transform.py:123-130."""
    ws = boxes[:, 2] - boxes[:, 0] + 1
    hs = boxes[:, 3] - boxes[:, 1] + 1
    return np.where((ws >= min_size) & (hs >= min_size))[0]


# ---- lib/pylayer/proposal_layer.py ---------------------------------------------------------------
def proposal_candidates(rpn_cls_prob, rpn_bbox_pred, im_info, feat_stride=16):
    """Steps 1-5 of ProposalLayer.forward, lib/pylayer/proposal_layer.py:75-145: fg scores are channels A..2A-1,
    anchors/deltas/scores ordered (h, w, a); decode all anchors, clip to im_info[:2], keep w,h >= 16*scale,
    argsort()[::-1][:6000].  Returns (proposals [n,4] f32, scores [n,1] f32) sorted by descending score."""
    anchors0 = generate_anchors()
    A = anchors0.shape[0]
    scores = rpn_cls_prob[:, A:, :, :]
    im_info = im_info[0, :]
    height, width = scores.shape[-2:]
    sx, sy = np.meshgrid(np.arange(0, width) * feat_stride, np.arange(0, height) * feat_stride)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    K = shifts.shape[0]
    anchors = (anchors0.reshape((1, A, 4)) + shifts.reshape((1, K, 4)).transpose((1, 0, 2))).reshape((K * A, 4))
    deltas = rpn_bbox_pred.transpose((0, 2, 3, 1)).reshape((-1, 4))
    scores = scores.transpose((0, 2, 3, 1)).reshape((-1, 1))
    proposals = bbox_transform_inv(anchors, deltas)
    proposals, _ = clip_boxes(proposals, im_info[:2])
    keep = filter_small_boxes(proposals, RPN_MIN_SIZE * im_info[2])
    proposals, scores = proposals[keep, :], scores[keep]
    # reference: scores.ravel().argsort()[::-1] -- numpy's default sort is unstable, so the order of EQUAL scores is
    # unspecified there.  The oracle pins it: score descending, index ascending.  Identical to the
    # reference whenever the scores are distinct (all committed fixtures are).
    order = np.argsort(-scores.ravel(), kind="stable")
    order = order[:RPN_PRE_NMS_TOP_N]
    return proposals[order, :], scores[order]


def proposal_forward(rpn_cls_prob, rpn_bbox_pred, im_info, nms_fn=None):
    """ProposalLayer.forward, lib/pylayer/proposal_layer.py:52-175 (TEST phase) -> rois [R<=300, 5] float32."""
    nms_fn = nms_fn or native.gpu_nms
    proposals, scores = proposal_candidates(rpn_cls_prob, rpn_bbox_pred, im_info)
    keep = nms_fn(np.hstack((proposals, scores)), RPN_NMS_THRESH)
    keep = keep[:RPN_POST_NMS_TOP_N]
    proposals = proposals[keep, :]
    batch = np.zeros((proposals.shape[0], 1), dtype=np.float32)
    return np.hstack((batch, proposals.astype(np.float32, copy=False))).astype(np.float32, copy=False)


# ---- lib/pylayer/mask_layer.py / stage_bridge_layer.py --------------------------------------------
def mask_layer_forward_test(mask_output):
    """MaskLayer.forward_test, lib/pylayer/mask_layer.py:95-102: [R,441] -> [R,1,21,21] float32."""
    return mask_output.reshape((mask_output.shape[0], 1, MASK_SIZE, MASK_SIZE)).astype(np.float32, copy=False)


def stage_bridge_forward_test(rois, bbox_pred, seg_cls_prob, im_info):
    """StageBridgeLayer.forward_test, lib/pylayer/stage_bridge_layer.py:237-255: decode all 21 class boxes, take
    the box of argmax(seg_cls_prob) INCLUDING background, clip; built in float64, stored as float32 (:79-80)."""
    all_rois = bbox_transform_inv(rois[:, 1:5], bbox_pred)
    score_max = seg_cls_prob.argmax(axis=1)
    out = np.zeros((rois.shape[0], 5))
    for i in range(len(score_max)):
        out[i, 1:5] = all_rois[i, 4 * score_max[i]:4 * (score_max[i] + 1)]
    out[:, 1:5], _ = clip_boxes(out[:, 1:5], im_info[0, :2])
    return out.astype(np.float32, copy=False)


# ---- lib/utils/blob.py + tools/demo.py -----------------------------------------------------------
def resize_bilinear_cv(im, fx, fy):
    """cv2.resize(im, None, None, fx, fy, INTER_LINEAR) for float32 HxWxC (lib/utils/blob.py:47-48).
    cv2 is absent (third-party, version unpinned by the reference); this restates OpenCV's published
    algorithm: dsize = round(src*f); src coordinate = (dst + 0.5)/f - 0.5, floor, clamp to [0, n-1] with
    zero fractional weight at the clamp; horizontal pass then vertical pass in float32."""
    im = np.asarray(im, dtype=np.float32)
    h, w = im.shape[:2]
    dh, dw = int(round(h * fy)), int(round(w * fx))

    def taps(n_dst, n_src, f):
        src = (np.arange(n_dst, dtype=np.float64) + 0.5) * (1.0 / f) - 0.5
        src = src.astype(np.float32)
        i0 = np.floor(src).astype(np.int64)
        frac = (src - i0).astype(np.float32)
        lo = i0 < 0
        frac[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_src - 1
        frac[hi], i0[hi] = 0.0, n_src - 1
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, frac

    x0, x1, ax = taps(dw, w, fx)
    y0, y1, ay = taps(dh, h, fy)
    ax = ax[None, :, None]
    ay = ay[:, None, None]
    rows = im[:, x0, :] * (np.float32(1.0) - ax) + im[:, x1, :] * ax
    return (rows[y0] * (np.float32(1.0) - ay) + rows[y1] * ay).astype(np.float32)



def resize_bilinear_cv_to(im, width, height):
    """cv2.resize(im, (width, height)) (INTER_LINEAR), 2-D or HxWxC float32: the dsize form used for masks
    (lib/utils/voc_eval.py:241, lib/transform/mask_transform.py:160).  OpenCV derives the factors from the sizes,
    inv_scale = dsize / ssize, and then proceeds as for explicit factors (resize_bilinear_cv above)."""
    im = np.asarray(im, dtype=np.float32)
    squeeze = im.ndim == 2
    if squeeze:
        im = im[:, :, None]
    h, w = im.shape[:2]
    fx, fy = float(width) / w, float(height) / h

    def taps(n_dst, n_src, f):
        src = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (1.0 / f) - 0.5).astype(np.float32)
        i0 = np.floor(src).astype(np.int64)
        frac = (src - i0).astype(np.float32)
        lo = i0 < 0
        frac[lo], i0[lo] = 0.0, 0
        hi = i0 >= n_src - 1
        frac[hi], i0[hi] = 0.0, n_src - 1
        return i0, np.minimum(i0 + 1, n_src - 1), frac

    x0, x1, ax = taps(int(width), w, fx)
    y0, y1, ay = taps(int(height), h, fy)
    ax, ay = ax[None, :, None], ay[:, None, None]
    rows = im[:, x0, :] * (np.float32(1.0) - ax) + im[:, x1, :] * ax
    out = (rows[y0] * (np.float32(1.0) - ay) + rows[y1] * ay).astype(np.float32)
    return out[:, :, 0] if squeeze else out


def prep_im_for_blob(im, pixel_means=PIXEL_MEANS, target_size=TEST_SCALE, max_size=MAX_SIZE):
    """lib/utils/blob.py:36-50: float32, subtract BGR means BEFORE resizing, scale = 600/min side capped so that
    round(scale*max side) <= 1000."""
    im = im.astype(np.float32, copy=True)
    im -= pixel_means          # float64 means: the subtraction runs in float64 and is rounded to float32 once
    smin, smax = np.min(im.shape[0:2]), np.max(im.shape[0:2])
    scale = float(target_size) / float(smin)
    if np.round(scale * smax) > max_size:
        scale = float(max_size) / float(smax)
    return resize_bilinear_cv(im, scale, scale), scale


def im_list_to_blob(ims):
    """lib/utils/blob.py:17-33: zero-padded NHWC stack -> NCHW float32."""
    shape = np.array([im.shape for im in ims]).max(axis=0)
    blob = np.zeros((len(ims), shape[0], shape[1], 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, 0:im.shape[0], 0:im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2))


def prepare_mnc_args(im):
    """tools/demo.py:54-76 == lib/caffeWrapper/TesterWrapper.py:262-284: (data [1,3,H,W], im_info [[H,W,scale]], scale)."""
    im_r, scale = prep_im_for_blob(im)
    data = np.ascontiguousarray(im_list_to_blob([im_r]), dtype=np.float32)
    im_info = np.array([[data.shape[2], data.shape[3], scale]], dtype=np.float32)
    return data, im_info, scale


def im_detect_tail(rois, masks, scores, rois_ext, masks_ext, scores_ext, im_scale, im_shape):
    """tools/demo.py:84-100 == TesterWrapper.py:244-260: un-scale rois, clip to the ORIGINAL image, concat stages."""
    s = np.float32(im_scale)  # original numpy-1.x value-based casting keeps float32 (see module docstring)
    b1, _ = clip_boxes(rois[:, 1:5] / s, im_shape)
    b2, _ = clip_boxes(rois_ext[:, 1:5] / s, im_shape)
    return (np.concatenate((b1, b2), axis=0), np.concatenate((masks, masks_ext), axis=0),
            np.concatenate((scores, scores_ext), axis=0))


# ---- lib/transform/mask_transform.py -------------------------------------------------------------
def mask_voting_candidates(boxes, scores, num_classes, max_per_image, nms_fn=None, overlaps_fn=None):
    """Host half of gpu_mask_voting, lib/transform/mask_transform.py:213-270: per-class NMS(0.3) keeping <= 100,
    global threshold = the max_per_image-th best kept score, then for every kept box >= threshold its candidate
    set {IoU >= 0.5 over all boxes} with class-score weights normalised by python's sequential sum() (float64 accumulation
    as under numpy 1.x, see below)."""
    nms_fn = nms_fn or native.gpu_nms
    overlaps_fn = overlaps_fn or native.bbox_overlaps
    sup_boxes, sup_scores, tobesort = [[]], [[]], []
    for i in range(1, num_classes):
        dets = np.hstack((boxes.astype(np.float32), scores[:, i:i + 1]))
        inds = nms_fn(dets, MASK_MERGE_NMS_THRESH)
        ind_boxes, ind_scores = boxes[inds], scores[inds, i]
        nk = min(len(ind_scores), max_per_image)
        sup_boxes.append(ind_boxes[0:nk, :])
        sup_scores.append(ind_scores[0:nk])
        tobesort.extend(ind_scores[0:nk])
    sorted_scores = np.sort(tobesort)[::-1]
    nk = min(len(sorted_scores), max_per_image)
    thresh = sorted_scores[nk - 1]
    cand_inds, cand_w, cand_start, cand_scores, class_bar = [], [], [], [], []
    for c in range(1, num_classes):
        cls_box, cls_score = sup_boxes[c], sup_scores[c]
        keep = np.where(cls_score >= thresh)[0]
        new_sup = cls_box[keep]
        for i in range(len(new_sup)):
            ov = overlaps_fn(boxes.astype(np.float64), new_sup[i, np.newaxis].astype(np.float64))
            cur = np.where(ov >= MASK_MERGE_IOU_THRESH)[0]
            cand_inds.extend(cur)
            w = scores[cur, c]
            # reference: `cur_weights / sum(cur_weights)` (mask_transform.py:266).  Under the numpy 1.x the reference ran on,
            # python's sum() starts from int 0 and 0 + np.float32 promotes to float64 (scalar-scalar promotion), so the
            # accumulation is sequential float64; the float32 array is then divided by that float64 SCALAR in float32
            # (value-based casting).  numpy 2 would accumulate in float32 -- pinned here to the original behaviour.
            w = w / np.float32(sum(w.astype(np.float64)))
            cand_w.extend(w)
            cand_start.append(len(cand_inds))
        cand_scores.extend(cls_score[keep])
        class_bar.append(len(cand_scores))
    return (np.array(cand_inds, dtype=np.int32), np.array(cand_start, dtype=np.int32),
            np.array(cand_w, dtype=np.float32), np.array(cand_scores, dtype=np.float32), class_bar)


def gpu_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height,
                    nms_fn=None, mv_fn=None, overlaps_fn=None):
    """gpu_mask_voting, lib/transform/mask_transform.py:213-286 -> (list_result_mask[20], list_result_box[20])."""
    mv_fn = mv_fn or native.mv
    inds, start, wts, cscores, class_bar = mask_voting_candidates(
        boxes, scores, num_classes, max_per_image, nms_fn, overlaps_fn)
    result_mask, result_box = mv_fn(boxes.astype(np.float32), masks, inds, start, wts, im_height, im_width)
    result_box = np.hstack((result_box, cscores[:, np.newaxis]))
    lb, lm = [], []
    for i in range(num_classes - 1):
        s = class_bar[i - 1] if i > 0 else 0
        e = class_bar[i]
        lb.append(result_box[s:e, :])
        lm.append(result_mask[s:e, :, :, :])
    return lm, lb
