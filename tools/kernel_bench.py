#!/usr/bin/env python3
"""Micro-benchmark of the two MFMA kernels through the C ABI (HIP-event timing via mnc_prof_*).

    python tools/kernel_bench.py conv [--reps 20]      the 13 conv3x3 shapes of the VGG-16 trunk at 600x1000
    python tools/kernel_bench.py convx3                the same on the bf16x3 kernel (MNC_CONVX3_TILE=CT,PR overrides the tile)
    python tools/kernel_bench.py convwino              the same on the Winograd F(2x2,3x3) fp32 kernel (MNC_WINO_ROWS=1|2|4)
    python tools/kernel_bench.py convwino4             the same on the fused Winograd F(4x4,3x3) fp32 kernel (csrc/conv_wino4.hip)
    python tools/kernel_bench.py fc   [--reps 20]      the FC shapes of one head stage at 300 RoIs
    python tools/kernel_bench.py fcx3                  the same on the bf16x3 kernel
    python tools/kernel_bench.py conv1x1 [--f16]       the 1x1 convolutions of the ResNet-50 C4 trunk at 800x1333: the GEMM kernel
                                                       (csrc/conv1x1.hip; MNC_CONV1X1_TILE=ct,pt) next to the general kernel
Environment knobs understood by the library (tuning aids): MNC_CONV_COT=1|2|4."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import Dev  # noqa: E402

CONV = [("conv1_2", 600, 1000, 64, 64), ("conv2_1", 300, 500, 64, 128), ("conv2_2", 300, 500, 128, 128),
        ("conv3_1", 150, 250, 128, 256), ("conv3_2", 150, 250, 256, 256), ("conv4_1", 75, 125, 256, 512),
        ("conv4_2", 75, 125, 512, 512), ("conv5_1", 38, 63, 512, 512)]
# ResNet-50 C4 at 800x1333 (pool1 200x334): (name, H, W, Cin, Cout, stride, residual, how many such layers)
C11 = [("res2a_2a", 200, 334, 64, 64, 1, False, 1), ("res2_2c", 200, 334, 64, 256, 1, True, 3), ("res2a_b1", 200, 334, 64, 256, 1, False, 1),
       ("res2_2a", 200, 334, 256, 64, 1, False, 2), ("res3a_2a", 200, 334, 256, 128, 2, False, 1),
       ("res3a_b1", 200, 334, 256, 512, 2, False, 1), ("res3_2c", 100, 167, 128, 512, 1, True, 4),
       ("res3_2a", 100, 167, 512, 128, 1, False, 3), ("res4a_2a", 100, 167, 512, 256, 2, False, 1),
       ("res4a_b1", 100, 167, 512, 1024, 2, False, 1), ("res4_2c", 50, 84, 256, 1024, 1, True, 6),
       ("res4_2a", 50, 84, 1024, 256, 1, False, 5)]
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
    ap.add_argument("what", choices=["conv", "convx3", "convf16", "convsw", "convwino", "convwino4", "fc", "fcx3", "fcf16", "conv1x1"])
    ap.add_argument("--f16", action="store_true", help="conv1x1: packed fp16 tensors (mnc_conv1x1_f16_pk) instead of fp32")
    ap.add_argument("--mode", default="f16", help="convsw: bf16x3 | f16 | bf16 (mnc_conv3x3_lowp, packed tensors)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--shape", default=None, help="fc / fcx3: one extra shape M,N,K (e.g. 40,4096,50176)")
    ap.add_argument("--relu-input", action="store_true", help="fc*: activations max(x, 0) (half zeros, as behind a ReLU) instead of N(0, 1): "
                    "the matrix pipe is power limited, operand values move the clock")
    ap.add_argument("--pair", action="store_true", help="fcx3 / fcf16: TWO products of the shape in one launch (mnc_fc_lowp_pair, activations "
                    "already stage-major: the box + mask branch call of the head stages)")
    ap.add_argument("--gap-ms", type=float, default=0.0, help="fc*: synchronise and sleep this long before every timed launch (a kernel that "
                    "starts on a rested chip runs at a higher clock than the same kernel back to back)")
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
    if args.what == "conv1x1":
        tot_new = tot_old = tot_fl = tot_by = 0.0
        for name, H, W, Cin, Cout, stride, residual, count in C11:
            if args.only and args.only not in name:
                continue
            OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
            xf = rng.normal(size=(Cin * H * W,)).astype(np.float32)
            wf = (rng.normal(size=(Cout * Cin,)) * 0.05).astype(np.float32)
            b = dev.put(np.zeros(Cout, np.float32))
            rf = rng.normal(size=(Cout * OH * OW,)).astype(np.float32)
            y = dev.empty((Cout * OH * OW,))
            raw = dev.put(wf)
            wn = dev.empty((Cin * ((Cout + 31) // 32) * 32,))
            dev.call("mnc_pack_conv1x1", raw, wn, Cout, Cin, 1 if args.f16 else 0)
            if args.f16:
                x = dev.put(xf.astype(np.float16), dtype=np.float16)
                r = dev.put(rf.astype(np.float16), dtype=np.float16) if residual else None
                new = lambda: dev.call("mnc_conv1x1_f16_pk", x, wn, b, r, y, H, W, Cin, Cout, stride, 1, 1, 1)
                wo = dev.empty((((Cin + 31) // 32) * 32 * Cout,))
                dev.call("mnc_pack_conv_weights_f16", raw, wo, Cout, Cin, 1, 1)
                x32, r32 = dev.put(xf), (dev.put(rf) if residual else None)
                old = lambda: dev.call("mnc_conv2d_f16", x32, wo, b, r32, y, H, W, Cin, Cout, 1, 1, stride, 0, 1)
                by = 2.0 * (Cin * OH * OW + Cout * OH * OW * (2 if residual else 1) + Cin * Cout)
            else:
                x = dev.put(xf)
                r = dev.put(rf) if residual else None
                new = lambda: dev.call("mnc_conv1x1", x, wn, b, r, y, H, W, Cin, Cout, stride, 1)
                wo = dev.empty((Cin * Cout,))
                dev.call("mnc_pack_conv_weights", raw, wo, Cout, Cin, 1, 1)
                old = lambda: dev.call("mnc_conv2d", x, wo, b, r, y, H, W, Cin, Cout, 1, 1, stride, 0, 1)
                by = 4.0 * (Cin * OH * OW + Cout * OH * OW * (2 if residual else 1) + Cin * Cout)
            res = {}
            for tag, fn in (("new", new), ("old", old)):
                for _ in range(3):
                    fn()
                dev.call("mnc_prof_reset")
                for _ in range(args.reps):
                    fn()
                res[tag] = np.array([t for n_, t in records(dev) if n_.startswith("conv")])
            fl = 2.0 * OH * OW * Cin * Cout
            tn, to = np.median(res["new"]), np.median(res["old"])
            print("%-9s %3dx%-3d %4d->%-4d s%d %s x%d  gemm %.1f us (min %.1f)  %.1f TF/s  %.2f TB/s | general %.1f us  %.1f TF/s" %
                  (name, H, W, Cin, Cout, stride, "res" if residual else "   ", count, 1e3 * tn, 1e3 * res["new"].min(),
                   fl / tn / 1e9, by / tn / 1e9, 1e3 * to, fl / to / 1e9), flush=True)
            tot_new += count * tn; tot_old += count * to; tot_fl += count * fl; tot_by += count * by
        if not args.only:
            print("all 1x1 layers of the trunk (%s): gemm %.3f ms = %.1f TF/s, %.2f TB/s algorithmic | general %.3f ms = %.1f TF/s"
                  % ("f16, packed tensors" if args.f16 else "fp32", tot_new, tot_fl / tot_new / 1e9, tot_by / tot_new / 1e9,
                     tot_old, tot_fl / tot_old / 1e9))
    elif args.what == "convsw":
        from mnc_amd import _lib
        m = {"bf16x3": 0, "f16": 1, "bf16": 2}[args.mode]
        for name, H, W, Cin, Cout in CONV:
            if args.only and args.only not in name:
                continue
            xf = dev.put(np.maximum(rng.normal(size=(Cin * H * W,)), 0).astype(np.float32) if args.relu_input else
                         rng.normal(size=(Cin * H * W,)).astype(np.float32))
            x = dev.empty((Cin * H * W,))
            dev.call("mnc_act_pack", xf, x, Cin * H * W, m)
            b = dev.put(np.zeros(Cout, np.float32))
            y = dev.empty((Cout * H * W,))
            raw = dev.put((rng.normal(size=(Cout * Cin * 9,)) * 0.05).astype(np.float32))
            w = dev.empty((_lib.load().mnc_conv3x3_lowp_weight_bytes(m, Cout, Cin) // 4,))
            dev.call("mnc_pack_conv3x3_lowp", m, raw, w, Cout, Cin)
            for _ in range(3):
                dev.call("mnc_conv3x3_lowp", m, x, w, b, y, None, H, W, Cin, Cout, 1)
            dev.call("mnc_prof_reset")
            for _ in range(args.reps):
                dev.call("mnc_conv3x3_lowp", m, x, w, b, y, None, H, W, Cin, Cout, 1)
            t = np.array([r[1] for r in records(dev) if r[0].startswith("conv3x3")])
            fl = 2.0 * H * W * 9 * Cin * Cout
            print("%-10s %4dx%-4d %3d->%-3d  med %.1f us  min %.1f us  %.1f TF/s (med)  %.1f TF/s (best)" %
                  (name, H, W, Cin, Cout, 1e3 * np.median(t), 1e3 * t.min(), fl / np.median(t) / 1e9, fl / t.min() / 1e9),
                  flush=True)
            mult = {"conv2_2": 1, "conv3_2": 2, "conv4_2": 2, "conv5_1": 4}.get(name, 1)
            total_ms += mult * np.median(t)
            total_fl += mult * fl
        if not args.only:
            print("trunk(12 layers)+rpn, %s: %.3f ms, %.1f TF/s" % (args.mode, total_ms, total_fl / total_ms / 1e9))
    elif args.what in ("conv", "convx3", "convf16", "convwino", "convwino4"):
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
                from mnc_amd import _lib
                w = dev.empty((_lib.load().mnc_conv3x3_lowp_weight_bytes(0 if mode == "bf16x3" else 1, Cout, Cin) // 4,))
                dev.call("mnc_pack_conv3x3_" + mode, raw, w, Cout, Cin)
                fn = "mnc_conv3x3_" + mode
                if args.packed:                       # 2-byte activations on both sides (buffers are large enough either way)
                    fn, extra = fn + "_pk", (1, 1)
            elif args.what == "convwino":
                raw = dev.put((rng.normal(size=(Cout * Cin * 9,)) * 0.05).astype(np.float32))
                w = dev.empty((Cin * Cout * 17,))
                dev.call("mnc_pack_conv3x3_wino", raw, w, Cout, Cin)
                fn = "mnc_conv3x3_wino"
            elif args.what == "convwino4":
                raw = dev.put((rng.normal(size=(Cout * Cin * 9,)) * 0.05).astype(np.float32))
                w = dev.empty((Cin * Cout * 36,))
                dev.call("mnc_pack_conv3x3_wino4", raw, w, Cout, Cin)
                fn = "mnc_conv3x3_wino4"
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
            af = rng.normal(size=(M * K,)).astype(np.float32)
            a = dev.put(np.maximum(af, 0) if args.relu_input else af)
            w = dev.put((rng.normal(size=(N * K,)) * 0.01).astype(np.float32))
            b = dev.put(np.zeros(N, np.float32))
            y = dev.empty((M * N,))
            fn = "mnc_fc"
            if args.what == "fcx3":
                wp = dev.empty(((N + 127) // 128 * 128 * K,))
                dev.call("mnc_pack_fc_bf16x3", w, wp, N, K)
                w, fn = wp, "mnc_fc_bf16x3"
            elif args.what == "fcf16":
                wp = dev.empty(((N + 127) // 128 * 128 * K // 2,))
                dev.call("mnc_pack_fc_f16", w, wp, N, K)
                w, fn = wp, "mnc_fc_f16"
            if args.pair and args.what in ("fcx3", "fcf16"):
                f16 = 1 if args.what == "fcf16" else 0
                asm = dev.empty((M * K // 2 if f16 else M * K,))
                dev.call("mnc_fc_pack_act", a, asm, M, K, f16)
                y1 = dev.empty((M * N,))
                mode = 1 if f16 else 0

                def run():
                    dev.call("mnc_fc_lowp_pair", mode, None, asm, None, asm, M, w, w, b, b, y, y1, M, N, K, N, 1, None, None, 0)
            else:
                def run():
                    dev.call(fn, a, w, b, y, M, N, K, N, 1)
            for _ in range(3):
                run()
            dev.call("mnc_prof_reset")
            for _ in range(args.reps):
                if args.gap_ms > 0:
                    dev.call("mnc_ctx_sync")
                    time.sleep(args.gap_ms * 1e-3)
                run()
            rec = records(dev)
            t = np.array([r[1] for r in rec if r[0].startswith("fc_mfma") or r[0] in ("fc_bf16x3", "fc_bf16x3_small", "fc_f16", "fc_f16_small")])
            tr = np.array([r[1] for r in rec if r[0] == "fc_reduce"] or [0.0])
            ts = np.array([r[1] for r in rec if r[0] in ("fc_bf16x3_split", "fc_f16_convert")] or [0.0])
            fl = 2.0 * M * N * K * (2 if args.pair and args.what in ("fcx3", "fcf16") else 1)
            tot = np.median(t) + np.median(tr) + np.median(ts)
            print("%-12s M=%d N=%-4d K=%-6d  gemm %.1f us + reduce %.1f us + split %.1f us = %.1f us   %.1f TF/s (gemm)  %.1f TF/s (all)" %
                  (name, M, N, K, 1e3 * np.median(t), 1e3 * np.median(tr), 1e3 * np.median(ts), 1e3 * tot,
                   fl / np.median(t) / 1e9, fl / tot / 1e9), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
