"""RPN anchors (reference: lib/transform/anchors.py:38-122).  The table is a compile-time constant of the model:
3 aspect ratios x 3 scales around the 16x16 window (0,0,15,15), in the reference's enumeration order (ratio-major).
NB the MATLAB table quoted in the reference's comment (anchors.py:10-35) is 1-based; the code's result is below."""
import numpy as np


def _centre(box):
    w, h = box[2] - box[0] + 1.0, box[3] - box[1] + 1.0
    return w, h, box[0] + 0.5 * (w - 1.0), box[1] + 0.5 * (h - 1.0)


def _around(cx, cy, ws, hs):
    ws, hs = np.asarray(ws, dtype=np.float64), np.asarray(hs, dtype=np.float64)
    return np.stack([cx - 0.5 * (ws - 1.0), cy - 0.5 * (hs - 1.0), cx + 0.5 * (ws - 1.0), cy + 0.5 * (hs - 1.0)], axis=1)


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    w, h, cx, cy = _centre(np.array([0.0, 0.0, base_size - 1.0, base_size - 1.0]))
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    rows = []
    for rbox in _around(cx, cy, ws, hs):
        rw, rh, rcx, rcy = _centre(rbox)
        rows.append(_around(rcx, rcy, rw * scales, rh * scales))
    return np.vstack(rows)


def generate_shifted_anchors(anchors, height, width, feat_stride):
    """All anchors of an height x width map, rows ordered (h, w, a) (anchors.py:105-122, proposal_layer.py:84-100)."""
    sx = np.arange(width) * feat_stride
    sy = np.arange(height) * feat_stride
    shift = np.stack(np.broadcast_arrays(sx[None, :], sy[:, None], sx[None, :], sy[:, None]), axis=-1)  # [H,W,4]
    return (shift[:, :, None, :] + anchors[None, None, :, :]).reshape(-1, 4)
