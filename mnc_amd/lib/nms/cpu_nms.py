"""`nms.cpu_nms.cpu_nms` -- the reference's CPU switch (cfg.USE_GPU_NMS = False, nms_wrapper.py:18-21).
NOT a fallback of the GPU path and not equivalent to it: the reference's Cython version suppresses on
`ovr >= thresh` (lib/nms/cpu_nms.pyx:65) while the GPU kernel uses a strict `>` (nms_kernel.cu:71)."""
import numpy as np


def cpu_nms(dets, thresh):
    dets = np.asarray(dets, dtype=np.float32)
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    suppressed = np.zeros(dets.shape[0], dtype=bool)
    keep = []
    for pos, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return keep
