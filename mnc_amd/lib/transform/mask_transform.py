"""`transform.mask_transform.gpu_mask_voting` (reference: lib/transform/mask_transform.py:213-286).

Host orchestration only: 20 per-class NMS calls (HIP), a global score threshold, one IoU row per surviving box, then a
single call into the fused HIP mask-voting kernels (nms.mv.mv).  The cv2-based cpu_mask_voting of the reference uses a
different interpolation and is not part of the hot path."""
import numpy as np

from mnc_config import cfg
from nms.gpu_nms import gpu_nms_batched
from nms.nms_wrapper import nms
from nms.mv import mv
from utils.cython_bbox import bbox_overlaps


def build_voting_candidates(boxes, scores, num_classes, max_per_image):
    """-> (candidate_inds i32[C], candidate_start i32[R] (END offsets), candidate_weights f32[C],
           candidate_scores f32[R], class_bar list[num_classes-1])."""
    boxes32 = boxes.astype(np.float32)
    kept = {}
    pool = []
    if cfg.USE_GPU_NMS and boxes32.shape[0] > 0:
        # the 20 per-class problems share the box set: one batched device call instead of 20 synchronous ones; only the
        # first max_per_image survivors of each class are used below, so the scan stops there (identical prefix)
        per_class = gpu_nms_batched(boxes32, scores[:, 1:num_classes], cfg.TEST.MASK_MERGE_NMS_THRESH,
                                    device_id=cfg.GPU_ID, max_keep=max_per_image)
    else:
        per_class = [nms(np.hstack((boxes32, scores[:, c:c + 1])), cfg.TEST.MASK_MERGE_NMS_THRESH)
                     for c in range(1, num_classes)]
    for c in range(1, num_classes):
        order = per_class[c - 1][:max_per_image]
        kept[c] = (boxes[order], scores[order, c])
        pool.extend(kept[c][1])
    if not pool:      # the reference would raise IndexError here (mask_transform.py:244); nothing to vote on
        z = np.zeros(0, np.int32)
        return z, z.copy(), np.zeros(0, np.float32), np.zeros(0, np.float32), [0] * (num_classes - 1)
    ranked = np.sort(pool)[::-1]
    thresh = ranked[min(len(ranked), max_per_image) - 1]
    boxes64 = boxes.astype(np.float64)
    inds, weights, ends, out_scores, class_bar = [], [], [], [], []
    for c in range(1, num_classes):
        cls_boxes, cls_scores = kept[c]
        sel = np.where(cls_scores >= thresh)[0]
        for b in cls_boxes[sel]:
            ov = bbox_overlaps(boxes64, b[np.newaxis].astype(np.float64))
            members = np.where(ov >= cfg.TEST.MASK_MERGE_IOU_THRESH)[0]
            w = scores[members, c]
            # python's sum() as the reference ran it (:266, numpy 1.x): sequential float64 accumulation (0 + np.float32 promoted
            # to float64), then a float32 division by that scalar (value-based casting)
            w = w / np.float32(sum(w.astype(np.float64)))
            inds.extend(members)
            weights.extend(w)
            ends.append(len(inds))
        out_scores.extend(cls_scores[sel])
        class_bar.append(len(out_scores))
    return (np.array(inds, dtype=np.int32), np.array(ends, dtype=np.int32), np.array(weights, dtype=np.float32),
            np.array(out_scores, dtype=np.float32), class_bar)


def mask_overlap(box1, box2, mask1, mask2):
    """Region IoU of two boolean masks that live inside different integer boxes (mask_transform.py:16-46; used by the
    mAP^r evaluation, utils/voc_eval.py)."""
    x1, y1 = max(box1[0], box2[0]), max(box1[1], box2[1])
    x2, y2 = min(box1[2], box2[2]), min(box1[3], box2[3])
    if x1 > x2 or y1 > y2:
        return 0
    w, h = x2 - x1 + 1, y2 - y1 + 1
    ya, xa = y1 - box1[1], x1 - box1[0]
    yb, xb = y1 - box2[1], x1 - box2[0]
    inter_a = mask1[ya: ya + h, xa: xa + w]
    inter_b = mask2[yb: yb + h, xb: xb + w]
    assert inter_a.shape == inter_b.shape
    inter = np.logical_and(inter_b, inter_a).sum()
    union = mask1.sum() + mask2.sum() - inter
    if union < 1.0:
        return 0
    return float(inter) / float(union)


def _fused_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height):
    """The whole of gpu_mask_voting in one C-ABI call (mnc_mask_voting): per-class score order, batched per-class NMS,
    candidate sets and the fused voting kernels, all on the device with no Python between the steps.  The per-class
    order is np.argsort(-scores[:, c], kind="stable") (ties in index order), computed by the library."""
    import ctypes
    from mnc_amd import _lib
    n = boxes.shape[0]
    S = masks.shape[3]
    B = num_classes - 1
    masks = np.ascontiguousarray(masks, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    cap = B * min(max_per_image, n)
    out_mask = np.zeros((cap, 1, S, S), dtype=np.float32)
    out_box = np.zeros((cap, 4), dtype=np.int32)
    out_score = np.zeros(cap, dtype=np.float32)
    counts = np.zeros(B, dtype=np.int32)
    R = ctypes.c_int(0)
    _lib.call("mnc_mask_voting", _lib.ptr(boxes), _lib.ptr(masks), _lib.ptr(scores), None, n, num_classes, S,
              int(max_per_image), float(cfg.TEST.MASK_MERGE_NMS_THRESH), float(cfg.TEST.MASK_MERGE_IOU_THRESH),
              int(im_height), int(im_width), _lib.ptr(out_mask), _lib.ptr(out_box), _lib.ptr(out_score),
              _lib.ptr(counts), ctypes.addressof(R), int(cfg.GPU_ID))
    return _split_results(out_mask, out_box, out_score, counts, R.value, B)


def _split_results(out_mask, out_box, out_score, counts, R, B):
    result_box = np.hstack((out_box[:R], out_score[:R, np.newaxis]))       # int32 | float32 -> float64, as the reference
    list_mask, list_box, lo = [], [], 0
    for c in range(B):
        hi = lo + int(counts[c])
        list_box.append(result_box[lo:hi, :])
        list_mask.append(out_mask[lo:hi])
        lo = hi
    return list_mask, list_box


def _device_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height):
    """gpu_mask_voting on the engine's own device-resident outputs (mnc_amd.devarray.DeviceArray, from Net.detect_tail):
    order, per-class NMS, threshold, result rows, candidate sets and voting as one asynchronous launch sequence on the net's
    stream (mnc_vote_instances); the only host contact is the copy of the final records."""
    blk = boxes._net.vote_instances(boxes, masks, scores, num_classes, max_per_image, im_width, im_height,
                                    cfg.TEST.MASK_MERGE_NMS_THRESH, cfg.TEST.MASK_MERGE_IOU_THRESH)
    return blk.lists()


def gpu_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height):
    """masks [n,1,S,S], boxes [n,4], scores [n,num_classes] -> (list_result_mask, list_result_box), one entry per
    foreground class; boxes rows are [x1, y1, x2, y2, score]."""
    from mnc_amd.devarray import DeviceArray
    if cfg.USE_GPU_NMS and all(isinstance(a, DeviceArray) for a in (masks, boxes, scores)) and boxes.shape[0] > 0:
        return _device_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height)
    masks, boxes, scores = (np.asarray(a) for a in (masks, boxes, scores))
    if (cfg.USE_GPU_NMS and boxes.dtype == np.float32 and scores.dtype == np.float32 and boxes.shape[0] > 0
            and boxes.shape[1] == 4 and masks.shape[2] == masks.shape[3]):
        return _fused_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height)
    # generic path (float64 boxes, CPU NMS, ...): the reference's step-by-step composition
    inds, ends, weights, out_scores, class_bar = build_voting_candidates(boxes, scores, num_classes, max_per_image)
    result_mask, result_box = mv(boxes.astype(np.float32), masks, inds, ends, weights, im_height, im_width,
                                 device_id=cfg.GPU_ID)
    result_box = np.hstack((result_box, out_scores[:, np.newaxis]))
    list_mask, list_box = [], []
    lo = 0
    for hi in class_bar:
        list_box.append(result_box[lo:hi, :])
        list_mask.append(result_mask[lo:hi, :, :, :])
        lo = hi
    return list_mask, list_box
