#!/usr/bin/env python3
"""Where the host time of gpu_mask_voting goes (600 instances of a 600x1000 image, the bench workload).
    MNC_MV_TIMING=1 python tools/voting_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _init_paths  # noqa: F401,E402
from transform import mask_transform as mt  # noqa: E402

rng = np.random.default_rng(0)
n = 600
ctr = rng.uniform(100, 900, (n, 2)); wh = rng.uniform(40, 300, (n, 2))
boxes = np.clip(np.hstack((ctr - wh / 2, ctr + wh / 2)), 0, [999, 599, 999, 599]).astype(np.float32)
masks = rng.uniform(0, 1, (n, 1, 21, 21)).astype(np.float32)
scores = rng.dirichlet(np.ones(21) * 0.3, n).astype(np.float32)
for it in range(5):
    t0 = time.perf_counter()
    order = np.empty((20, n), np.int32)
    for c in range(20):
        order[c] = np.argsort(-scores[:, c + 1], kind="stable")
    t1 = time.perf_counter()
    lm, lb = mt.gpu_mask_voting(masks, boxes, scores, 21, 100, 1000, 600)
    t2 = time.perf_counter()
    print("argsort x20: %.0f us   gpu_mask_voting total: %.0f us   (%d results)" %
          (1e6 * (t1 - t0), 1e6 * (t2 - t1), sum(len(b) for b in lb)))
