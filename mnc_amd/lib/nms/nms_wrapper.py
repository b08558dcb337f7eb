"""`nms.nms_wrapper` -- dispatch with the reference's semantics (lib/nms/nms_wrapper.py:13-71)."""
from mnc_config import cfg
from nms.gpu_nms import gpu_nms
from nms.cpu_nms import cpu_nms


def nms(dets, thresh):
    """Indices (into `dets`) kept by greedy NMS.  Empty input -> [] (nms_wrapper.py:16-17)."""
    if dets.shape[0] == 0:
        return []
    if cfg.USE_GPU_NMS:
        return gpu_nms(dets, thresh, device_id=cfg.GPU_ID)
    return cpu_nms(dets, thresh)


def apply_nms(all_boxes, thresh):
    """Per class / per image NMS over a [class][image] table of (n,5) arrays (nms_wrapper.py:24-43)."""
    out = [[[] for _ in row] for row in all_boxes]
    for c, row in enumerate(all_boxes):
        for i, dets in enumerate(row):
            if len(dets) == 0:
                continue
            keep = nms(dets, thresh)
            if len(keep):
                out[c][i] = dets[keep, :].copy()
    return out


def apply_nms_mask_single(box, mask, thresh):
    if len(box) == 0:
        return box, mask
    keep = nms(box, thresh)
    if len(keep) == 0:
        return box, mask
    return box[keep, :].copy(), mask[keep, :].copy()
