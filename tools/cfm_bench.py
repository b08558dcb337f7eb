#!/usr/bin/env python3
"""Time the CFM test path (models/VGG16/cfm/test.prototxt with experiments/cfgs/VGG16/cfm.yml's test settings:
5-level pyramid 480..1024 capped at 1500, levels grouped 3 + 2 per forward, 2000 MCG proposals, chunks of 2000 / 500 rois) on
one synthetic 375x500 image with seeded synthetic weights and proposals.  Prints per-image wall time and the per-kernel
breakdown (HIP events).  Not the headline bench (bench.py): a measurement of the widened row.

    python tools/cfm_bench.py [--proposals 2000] [--iters 3] [--math fp32|bf16x3]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

import _init_paths  # noqa: F401
from mnc_amd import models, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--math", default=os.environ.get("MNC_MATH", "fp32"))
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=500)
    args = ap.parse_args()
    os.environ["MNC_MATH"] = args.math
    import scipy.io
    from caffeWrapper.TesterWrapper import TesterWrapper
    from mnc_config import cfg
    cfg.TEST.SCALES, cfg.TEST.MAX_SIZE = [480, 576, 688, 864, 1024], 1500
    cfg.TEST.GROUP_SCALE, cfg.TEST.MAX_ROIS_GPU, cfg.TEST.USE_TOP_K_MCG = 3, [2000, 500], 2000
    rng = np.random.default_rng(0)
    H, W, n = args.height, args.width, args.proposals
    with tempfile.TemporaryDirectory() as root:
        cfg.ROOT_DIR = root
        cfg.TEST.MCG_MASKDB_DIR = os.path.join(root, "maskdb")
        os.makedirs(cfg.TEST.MCG_MASKDB_DIR)
        np.save(os.path.join(root, "im0.npy"), rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
        # MCG-like size mix: log-uniform sides from 16 px to the whole image
        w = np.exp(rng.uniform(np.log(16), np.log(W - 2), n)).astype(np.int64)
        h = np.exp(rng.uniform(np.log(16), np.log(H - 2), n)).astype(np.int64)
        x1, y1 = (rng.uniform(0, 1, n) * (W - w)).astype(np.int64), (rng.uniform(0, 1, n) * (H - h)).astype(np.int64)
        boxes = np.stack([x1, y1, x1 + w - 1, y1 + h - 1], 1).astype(np.float64)
        yy, xx = np.mgrid[0:21, 0:21]
        masks = ((xx[None] - rng.uniform(6, 14, n)[:, None, None]) ** 2 + (yy[None] - rng.uniform(6, 14, n)[:, None, None]) ** 2
                 <= rng.uniform(4, 11, n)[:, None, None] ** 2)
        scipy.io.savemat(os.path.join(cfg.TEST.MCG_MASKDB_DIR, "im0.mat"), {"boxes": boxes, "masks": masks})

        class Imdb(object):
            name, image_index, _image_index, num_classes = "cfm_bench", ["im0"], ["im0"], 21

            def image_path_at(self, i):
                return os.path.join(root, "im0.npy")

        path = models.write_cfm_test_prototxt()
        t0 = time.time()
        weights = synth.synthetic_weights(path, seed=0)
        t = TesterWrapper(path, Imdb(), weights, "cfm")
        print("net ready in %.1f s" % (time.time() - t0), file=sys.stderr)
        calls = []
        real = t.net.forward

        def spy(**kw):
            calls.append((kw.get("start"), tuple(kw["data"].shape) if "data" in kw else None, len(kw["rois"])))
            return real(**kw)
        t.net.forward = spy
        t.cfm_network_forward(0)                                      # warm-up: weight packing, buffer growth
        plan = list(calls)
        t.net.forward = real
        times = []
        for _ in range(args.iters):
            t.net.sync()
            t0 = time.perf_counter()
            t.cfm_network_forward(0)
            t.net.sync()
            times.append(time.perf_counter() - t0)
        t.net.profile(1)
        t.cfm_network_forward(0)
        t.net.sync()
        agg = {}
        for name, ms, fl, by in t.net.profile_records():
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += fl
        t.net.profile(0)
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        dev_ms = sum(v[1] for v in agg.values())
        print(json.dumps({"workload": "cfm vgg16 %dx%d, %d proposals, scales 480-1024 (3+2 levels/forward)" % (H, W, n),
                          "math": args.math, "forwards": [{"start": c[0], "data": c[1], "rois": c[2]} for c in plan],
                          "ms_per_image_wall": round(min(times) * 1e3, 2), "ms_per_image_kernels": round(dev_ms, 2),
                          "kernels": [{"name": k, "calls": v[0], "ms": round(v[1], 3),
                                       "tflops": round(v[2] / v[1] / 1e9, 1) if v[1] > 0 and v[2] > 0 else None}
                                      for k, v in rows[:14]]}))
        t.net.close()


if __name__ == "__main__":
    main()
