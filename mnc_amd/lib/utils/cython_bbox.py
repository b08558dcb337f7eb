"""`utils.cython_bbox.bbox_overlaps` -- same call as the reference's Cython extension (lib/utils/bbox.pyx:15-55),
served by the C function mnc_bbox_overlaps of libmnc_hip.so (a host function there as well)."""
import numpy as np

from mnc_amd import _lib


def bbox_overlaps(boxes, query_boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64)
    if boxes.ndim != 2 or query_boxes.ndim != 2 or boxes.shape[1] < 4 or query_boxes.shape[1] < 4:
        raise ValueError("bbox_overlaps expects (N,4) and (K,4) float arrays")
    if boxes.shape[1] != 4:
        boxes = np.ascontiguousarray(boxes[:, :4])
    if query_boxes.shape[1] != 4:
        query_boxes = np.ascontiguousarray(query_boxes[:, :4])
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((n, k), dtype=np.float64)
    _lib.call("mnc_bbox_overlaps", _lib.ptr(boxes), n, _lib.ptr(query_boxes), k, _lib.ptr(out))
    return out
