"""`transform.mask_transform.gpu_mask_voting` (reference: lib/transform/mask_transform.py:213-286).

Host orchestration only: 20 per-class NMS calls (HIP), a global score threshold, one IoU row per surviving box, then a
single call into the fused HIP mask-voting kernels (nms.mv.mv).  The cv2-based cpu_mask_voting of the reference uses a
different interpolation and is not part of the hot path."""
import numpy as np

from mnc_config import cfg
from nms.nms_wrapper import nms
from nms.mv import mv
from utils.cython_bbox import bbox_overlaps


def build_voting_candidates(boxes, scores, num_classes, max_per_image):
    """-> (candidate_inds i32[C], candidate_start i32[R] (END offsets), candidate_weights f32[C],
           candidate_scores f32[R], class_bar list[num_classes-1])."""
    boxes32 = boxes.astype(np.float32)
    kept = {}
    pool = []
    for c in range(1, num_classes):
        order = nms(np.hstack((boxes32, scores[:, c:c + 1])), cfg.TEST.MASK_MERGE_NMS_THRESH)[:max_per_image]
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
            w = w / sum(w)          # python's sequential float32 sum, as in the reference (:266)
            inds.extend(members)
            weights.extend(w)
            ends.append(len(inds))
        out_scores.extend(cls_scores[sel])
        class_bar.append(len(out_scores))
    return (np.array(inds, dtype=np.int32), np.array(ends, dtype=np.int32), np.array(weights, dtype=np.float32),
            np.array(out_scores, dtype=np.float32), class_bar)


def gpu_mask_voting(masks, boxes, scores, num_classes, max_per_image, im_width, im_height):
    """masks [n,1,S,S], boxes [n,4], scores [n,num_classes] -> (list_result_mask, list_result_box), one entry per
    foreground class; boxes rows are [x1, y1, x2, y2, score]."""
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
