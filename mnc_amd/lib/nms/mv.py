"""`nms.mv.mv` -- same call as the reference's Cython wrapper (lib/nms/gpu_mv.pyx:13-31; the extension is named
`nms.mv`, lib/setup.py:146) over the fused HIP kernels behind mnc_mv (mnc_amd/csrc/mv.hip).  Unlike the reference,
an empty result list returns empty arrays instead of raising IndexError, and device_id is honoured."""
import numpy as np

from mnc_amd import _lib


def mv(all_boxes, all_masks, candidate_inds, candidate_start, candidate_weights, image_height, image_width,
       device_id=0):
    all_boxes = np.ascontiguousarray(all_boxes, dtype=np.float32)
    all_masks = np.ascontiguousarray(all_masks, dtype=np.float32)
    candidate_inds = np.ascontiguousarray(candidate_inds, dtype=np.int32)
    candidate_start = np.ascontiguousarray(candidate_start, dtype=np.int32)
    candidate_weights = np.ascontiguousarray(candidate_weights, dtype=np.float32)
    if all_boxes.ndim != 2 or all_masks.ndim != 4:
        raise ValueError("mv expects boxes (N, box_dim) and masks (N, 1, S, S)")
    n, box_dim = all_boxes.shape
    mask_size = all_masks.shape[3]
    result_num = candidate_start.shape[0]
    result_mask = np.zeros((result_num, 1, all_masks.shape[2], mask_size), dtype=np.float32)
    result_box = np.zeros((result_num, box_dim), dtype=np.int32)
    if result_num == 0:
        return result_mask, result_box
    if box_dim != 4:
        raise ValueError("mv: box_dim must be 4 (gpu_mv.pyx allocates result_box with box_dim columns)")
    _lib.call("mnc_mv", _lib.ptr(all_boxes), _lib.ptr(all_masks), n, _lib.ptr(candidate_inds),
              _lib.ptr(candidate_start), _lib.ptr(candidate_weights), candidate_inds.shape[0], int(image_height),
              int(image_width), box_dim, mask_size, result_num, _lib.ptr(result_mask), _lib.ptr(result_box),
              int(device_id))
    return result_mask, result_box
