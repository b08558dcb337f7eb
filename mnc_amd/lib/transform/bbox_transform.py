"""Box decoding / clipping used on the inference path (reference: lib/transform/bbox_transform.py:64-130).
Arithmetic follows the reference operation by operation in the dtype of `deltas` (float32 on this path), so the
results are bit-identical to it."""
import numpy as np


def bbox_transform_inv(boxes, deltas):
    """Apply (dx, dy, dw, dh) deltas; `deltas` may hold K classes per row as [N, 4K]."""
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    boxes = boxes.astype(deltas.dtype, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    w, h, cx, cy = w[:, None], h[:, None], cx[:, None], cy[:, None]
    pcx = deltas[:, 0::4] * w + cx
    pcy = deltas[:, 1::4] * h + cy
    pw = np.exp(deltas[:, 2::4]) * w
    ph = np.exp(deltas[:, 3::4]) * h
    out = np.empty(deltas.shape, dtype=deltas.dtype)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw        # no "-1" on the far edge (bbox_transform.py:94-97)
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape):
    """Clamp x to [0, W-1], y to [0, H-1]; also returns the row indices that were already inside."""
    xmax, ymax = im_shape[1] - 1, im_shape[0] - 1
    x1, y1, x2, y2 = boxes[:, 0::4], boxes[:, 1::4], boxes[:, 2::4], boxes[:, 3::4]
    inside = np.where((x1 >= 0) & (x2 <= xmax) & (y1 >= 0) & (y2 <= ymax))[0]
    out = np.empty(boxes.shape, dtype=boxes.dtype)
    out[:, 0::4] = np.maximum(np.minimum(x1, xmax), 0)
    out[:, 1::4] = np.maximum(np.minimum(y1, ymax), 0)
    out[:, 2::4] = np.maximum(np.minimum(x2, xmax), 0)
    out[:, 3::4] = np.maximum(np.minimum(y2, ymax), 0)
    return out, inside


def filter_small_boxes(boxes, min_size):
    """Indices of boxes whose +1 width and height are both >= min_size."""
    return np.where(((boxes[:, 2] - boxes[:, 0] + 1) >= min_size) & ((boxes[:, 3] - boxes[:, 1] + 1) >= min_size))[0]
