"""`nms.gpu_nms.gpu_nms` -- same call as the reference's Cython wrapper (lib/nms/gpu_nms.pyx:16-31) over the HIP
kernel behind mnc_nms (mnc_amd/csrc/nms.hip).  The score sort stays on the host as in the reference; where the
reference's `argsort()[::-1]` leaves the order of EQUAL scores to numpy's unstable sort, this wrapper defines it (score
descending, index ascending) so that every path of this package -- host or device -- agrees.  Distinct scores: identical."""
import ctypes

import numpy as np

from mnc_amd import _lib


def gpu_nms(dets, thresh, device_id=0, max_keep=-1):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] < 5:
        raise ValueError("gpu_nms expects an (n, 5) float32 array")
    n, dim = dets.shape
    if n == 0:
        return []
    order = np.argsort(-dets[:, 4], kind="stable")      # score descending, ties by ascending index (see module doc)
    sorted_dets = np.ascontiguousarray(dets[order, :])
    keep = np.zeros(n, dtype=np.int32)
    num = ctypes.c_int(0)
    if max_keep is None or max_keep < 0:
        _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(sorted_dets), n, dim, float(thresh),
                  int(device_id))
    else:
        _lib.call("mnc_nms_topk", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(sorted_dets), n, dim, float(thresh),
                  int(max_keep), int(device_id))
    return [int(i) for i in order[keep[:num.value]]]


def gpu_nms_batched(boxes, scores, thresh, device_id=0, max_keep=-1):
    """NMS of ONE box set under several score columns at once -- gpu_mask_voting's per-class loop
    (lib/transform/mask_transform.py:228-240) as a single device round trip (mnc_nms_batched).

    boxes [n,4] float32, scores [n,B] -> list of B keep lists; entry b equals gpu_nms(hstack(boxes, scores[:, b]))
    (truncated to max_keep when given): the per-column order is the same `argsort()[::-1]` the per-call wrapper uses."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n, batch = scores.shape
    if n == 0 or batch == 0:
        return [[] for _ in range(batch)]
    orders = np.empty((batch, n), dtype=np.int32)
    for b in range(batch):
        orders[b] = np.argsort(-scores[:, b], kind="stable")
    keep = np.zeros((batch, n), dtype=np.int32)
    num = np.zeros(batch, dtype=np.int32)
    _lib.call("mnc_nms_batched", _lib.ptr(keep), _lib.ptr(num), _lib.ptr(boxes), n, boxes.shape[1], _lib.ptr(orders),
              batch, float(thresh), int(-1 if max_keep is None else max_keep), int(device_id))
    return [[int(i) for i in orders[b][keep[b, :num[b]]]] for b in range(batch)]
