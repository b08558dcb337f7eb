#!/usr/bin/env python3
"""Time the per-image body of `tools/test_net.py` on one synthetic VOC-sized image, host work
included (image preparation, forward, result hand-over) -- what bench.py's HBM-resident headline leaves out:

  --task seg   TesterWrapper._segmentation_forward + gpu_mask_voting of the 5-stage MNC graph (375x500 -> 600x800)
  --task cfm   TesterWrapper.cfm_network_forward with experiments/cfgs/VGG16/cfm.yml's test settings: 5-level pyramid
               480..1024 capped at 1500, levels grouped 3 + 2 per forward, 2000 MCG proposals in chunks of 2000 / 500 rois

  --task resnet  the same seg body on the ResNet-50 C4 trunk graph (models.mnc_resnet50_test_prototxt; BASELINE configs[4]):
               800x1333 image, 1000 proposals per stage (row n4 -- first correct path, general-convolution kernels untuned)

Seeded synthetic weights, pixels and proposals.  Prints wall time per image and the per-kernel breakdown (HIP events).

    python tools/task_bench.py --task seg|cfm|resnet [--iters 5] [--math fp32|bf16x3|f16] [--host-prep]
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
    ap.add_argument("--task", default="cfm", choices=["seg", "cfm", "resnet"])
    ap.add_argument("--host-prep", action="store_true", help="numpy image preparation (cfg.TEST.DEVICE_PREP = False)")
    ap.add_argument("--proposals", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--math", default=os.environ.get("MNC_MATH", "fp32"))
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    args = ap.parse_args()
    resnet = args.task == "resnet"
    if resnet:
        args.task = "seg"
    args.height = args.height or (800 if resnet else 375)
    args.width = args.width or (1333 if resnet else 500)
    os.environ["MNC_MATH"] = args.math
    import scipy.io
    from caffeWrapper.TesterWrapper import TesterWrapper
    from mnc_config import cfg
    cfg.TEST.DEVICE_PREP = not args.host_prep
    if resnet:
        cfg.TEST.SCALES, cfg.TRAIN.MAX_SIZE, cfg.TEST.RPN_POST_NMS_TOP_N = (800,), 1333, 1000
    if args.task == "cfm":
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

        path = (models.write_cfm_test_prototxt() if args.task == "cfm" else
                models.write_mnc_resnet50_test_prototxt() if resnet else models.write_mnc_5stage_test_prototxt())
        t0 = time.time()
        weights = synth.synthetic_weights(path, seed=0)
        t = TesterWrapper(path, Imdb(), weights, args.task)
        print("net ready in %.1f s" % (time.time() - t0), file=sys.stderr)
        from transform.mask_transform import gpu_mask_voting
        from utils.image_io import imread

        def body():
            if args.task == "cfm":
                return t.cfm_network_forward(0)
            im = imread(Imdb().image_path_at(0))
            masks, bxs, scores = t._segmentation_forward(im)
            return gpu_mask_voting(masks, bxs, scores, 21, 100, im.shape[1], im.shape[0])

        calls = []
        real = t.net.forward

        def spy(**kw):
            calls.append((kw.get("start"), tuple(kw["data"].shape) if "data" in kw else None,
                          len(kw["rois"]) if "rois" in kw else None))
            return real(**kw)
        t.net.forward = spy
        body()                                                        # warm-up: weight packing, buffer growth
        plan = list(calls)
        t.net.forward = real
        times = []
        for _ in range(args.iters):
            t.net.sync()
            t0 = time.perf_counter()
            body()
            t.net.sync()
            times.append(time.perf_counter() - t0)
        t.net.profile(1)
        body()
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
        what = ("cfm vgg16 %dx%d, %d proposals, scales 480-1024 (3+2 levels/forward)" % (H, W, n) if args.task == "cfm" else
                "mnc 5-stage %s %dx%d image -> %s, %d rois/stage, mask voting"
                % ("resnet50-c4" if resnet else "vgg16", H, W, "x".join(map(str, plan[0][1][2:])), cfg.TEST.RPN_POST_NMS_TOP_N))
        print(json.dumps({"workload": what, "math": args.math, "image_prep": "host" if args.host_prep else "device", "forwards": [{"start": c[0], "data": c[1], "rois": c[2]} for c in plan],
                          "ms_per_image_wall": round(min(times) * 1e3, 2), "ms_per_image_wall_median": round(sorted(times)[len(times) // 2] * 1e3, 2), "ms_per_image_kernels": round(dev_ms, 2),
                          "kernels": [{"name": k, "calls": v[0], "ms": round(v[1], 3),
                                       "tflops": round(v[2] / v[1] / 1e9, 1) if v[1] > 0 and v[2] > 0 else None}
                                      for k, v in rows[:14]]}))
        t.net.close()


if __name__ == "__main__":
    main()
