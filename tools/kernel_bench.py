#!/usr/bin/env python3
"""Micro-benchmark of the two MFMA kernels through the C ABI (HIP-event timing via mnc_prof_*).

    python tools/kernel_bench.py conv [--reps 20]      the 13 conv3x3 shapes of the VGG-16 trunk at 600x1000
    python tools/kernel_bench.py convx3                the same on the bf16x3 kernel (MNC_CONVX3_TILE=CT,PR overrides the tile)
    python tools/kernel_bench.py convwino              the same on the Winograd F(2x2,3x3) fp32 kernel (MNC_WINO_ROWS=1|2|4)
    python tools/kernel_bench.py fc   [--reps 20]      the FC shapes of one head stage at 300 RoIs
    python tools/kernel_bench.py fcx3                  the same on the bf16x3 kernel
Environment knobs understood by the library (tuning aids): MNC_CONV_COT=1|2|4."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import Dev  # noqa: E402

CONV = [("conv1_2", 600, 1000, 64, 64), ("conv2_1", 300, 500, 64, 128), ("conv2_2", 300, 500, 128, 128),
        ("conv3_1", 150, 250, 128, 256), ("conv3_2", 150, 250, 256, 256), ("conv4_1", 75, 125, 256, 512),
        ("conv4_2", 75, 125, 512, 512), ("conv5_1", 38, 63, 512, 512)]
FC = [("fc6_maskest", 300, 256, 100352), ("mask_pred", 300, 441, 256), ("fc6", 300, 4096, 25088), ("fc7", 300, 4096, 4096),
      ("heads", 300, 126, 8192)]


def records(dev):
    n = ctypes.c_int(0)
    dev.call("mnc_prof_count", ctypes.addressof(n))
    out = []
    name = ctypes.create_string_buffer(64)
    ms = ctypes.c_float(0)
    for i in range(n.value):
        dev.call("mnc_prof_get", i, ctypes.addressof(name), 64, ctypes.addressof(ms), None, None)
        out.append((name.value.decode(), ms.value))
    dev.call("mnc_prof_reset")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["conv", "convx3", "convf16", "convwino", "fc", "fcx3"])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--shape", default=None, help="fc / fcx3: one extra shape M,N,K (e.g. 40,4096,50176)")
    ap.add_argument("--packed", action="store_true", help="convx3 / convf16: 2-byte activation tensors in and out")
    ap.add_argument("--conv-shape", action="append", default=[], help="conv*: replace the layer list by H,W,Cin,Cout (repeatable)")
    args = ap.parse_args()
    if args.conv_shape:
        CONV[:] = [("custom%d" % i,) + tuple(int(v) for v in sh.split(",")) for i, sh in enumerate(args.conv_shape)]
    if args.shape:
        m, n, k = (int(v) for v in args.shape.split(","))
        FC[:] = [("custom", m, n, k)]
    dev = Dev(0)
    rng = np.random.default_rng(0)
    dev.call("mnc_prof_enable", 1)
    total_ms, total_fl = 0.0, 0.0
    if args.what in ("conv", "convx3", "convf16", "convwino"):
        for name, H, W, Cin, Cout in CONV:
            if args.only and args.only not in name:
                continue
            x = dev.put(rng.normal(size=(Cin * H * W,)).astype(np.float32))
            b = dev.put(np.zeros(Cout, np.float32))
            y = dev.empty((Cout * H * W,))
            fn = "mnc_conv3x3"
            extra = ()
            if args.what in ("convx3", "convf16"):
                mode = "bf16x3" if args.what == "convx3" else "f16"
                raw = dev.put((rng.normal(size=(Cout * Cin * 9,)) * 0.05).astype(np.float32))
                w = dev.empty(((Cin // 8) * Cout * 84,))
                dev.call("mnc_pack_conv3x3_" + mode, raw, w, Cout, Cin)
                fn = "mnc_conv3x3_" + mode
                if args.packed:                       # 2-byte activations on both sides (buffers are large enough either way)
                    fn, extra = fn + "_pk", (1, 1)
            elif args.what == "convwino":
                raw = dev.put((rng.normal(size=(Cout * Cin * 9,)) * 0.05).astype(np.float32))
                w = dev.empty((Cin * Cout * 17,))
                dev.call("mnc_pack_conv3x3_wino", raw, w, Cout, Cin)
                fn = "mnc_conv3x3_wino"
            else:
                w = dev.put((rng.normal(size=((Cin // 8) * Cout * 76,)) * 0.05).astype(np.float32))
            for _ in range(3):
                dev.call(fn, x, w, b, y, H, W, Cin, Cout, 1, *extra)
            dev.call("mnc_prof_reset")
            for _ in range(args.reps):
                dev.call(fn, x, w, b, y, H, W, Cin, Cout, 1, *extra)
            t = np.array([r[1] for r in records(dev) if r[0].startswith("conv3x3")])
            fl = 2.0 * H * W * 9 * Cin * Cout
            print("%-10s %4dx%-4d %3d->%-3d  med %.1f us  min %.1f us  %.1f TF/s (med)  %.1f TF/s (best)" %
                  (name, H, W, Cin, Cout, 1e3 * np.median(t), 1e3 * t.min(), fl / np.median(t) / 1e9, fl / t.min() / 1e9),
                  flush=True)
            mult = {"conv2_2": 1, "conv3_2": 2, "conv4_2": 2, "conv5_1": 4}.get(name, 1)
            total_ms += mult * np.median(t)
            total_fl += mult * fl
        if not args.only:
            print("trunk(12 layers)+rpn: %.3f ms, %.1f TF/s" % (total_ms, total_fl / total_ms / 1e9))
    else:
        for name, M, N, K in FC:
            if args.only and args.only not in name:
                continue
            a = dev.put(rng.normal(size=(M * K,)).astype(np.float32))
            w = dev.put((rng.normal(size=(N * K,)) * 0.01).astype(np.float32))
            b = dev.put(np.zeros(N, np.float32))
            y = dev.empty((M * N,))
            fn = "mnc_fc"
            if args.what == "fcx3":
                wp = dev.empty(((N + 127) // 128 * 128 * K,))
                dev.call("mnc_pack_fc_bf16x3", w, wp, N, K)
                w, fn = wp, "mnc_fc_bf16x3"
            for _ in range(3):
                dev.call(fn, a, w, b, y, M, N, K, N, 1)
            dev.call("mnc_prof_reset")
            for _ in range(args.reps):
                dev.call(fn, a, w, b, y, M, N, K, N, 1)
            rec = records(dev)
            t = np.array([r[1] for r in rec if r[0].startswith("fc_mfma") or r[0] in ("fc_bf16x3", "fc_bf16x3_small")])
            tr = np.array([r[1] for r in rec if r[0] == "fc_reduce"] or [0.0])
            ts = np.array([r[1] for r in rec if r[0] == "fc_bf16x3_split"] or [0.0])
            fl = 2.0 * M * N * K
            tot = np.median(t) + np.median(tr) + np.median(ts)
            print("%-12s M=%d N=%-4d K=%-6d  gemm %.1f us + reduce %.1f us + split %.1f us = %.1f us   %.1f TF/s (gemm)  %.1f TF/s (all)" %
                  (name, M, N, K, 1e3 * np.median(t), 1e3 * np.median(tr), 1e3 * np.median(ts), 1e3 * tot,
                   fl / np.median(t) / 1e9, fl / tot / 1e9), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
