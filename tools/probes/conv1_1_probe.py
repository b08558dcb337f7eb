#!/usr/bin/env python3
"""conv1_1 (3 -> 64 channels, 600x1000) back to back: the matrix-pipe kernel and the VALU kernel (MNC_CONV_COT=-1), per-launch HIP
event times -- is the 60 us it takes inside an image the kernel, or the clock ramp behind the upload / prep lull in front of it?"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import Dev  # noqa: E402

def records(dev):
    n = ctypes.c_int(0)
    dev.call("mnc_prof_count", ctypes.addressof(n))
    out = []
    name = ctypes.create_string_buffer(64); ms = ctypes.c_float(0)
    for i in range(n.value):
        dev.call("mnc_prof_get", i, ctypes.addressof(name), 64, ctypes.addressof(ms), None, None)
        out.append(ms.value * 1e3)
    dev.call("mnc_prof_reset")
    return out

dev = Dev(0)
H, W, C = 600, 1000, 64
rng = np.random.default_rng(0)
x = rng.normal(size=(3, H, W)).astype(np.float32)
w = (rng.normal(size=(C, 3, 3, 3)) * 0.2).astype(np.float32)
b = rng.normal(size=C).astype(np.float32)
d_x, d_w, d_b = dev.put(x), dev.put(w), dev.put(b)
d_y = dev.empty((C * H * W,))
for cot, cap in ((None, None), (None, "512"), (None, "768"), (None, "1024"), (None, "1536"), (None, "3072"), (None, "4096"), ("-1", None)):
    dev.tune("CONV_COT", cot)
    dev.tune("CONV_ROWS", cap)
    dev.call("mnc_prof_enable", 1)
    for _ in range(30):
        dev.call("mnc_conv3x3_c3", d_x, d_w, d_b, d_y, H, W, C, 1)
    dev.sync()
    t = records(dev)
    dev.call("mnc_prof_enable", 0)
    print("grid cap %s CONV_COT=%s: first %.1f us, median %.1f us, min %.1f us" % (cap, cot, t[0], sorted(t)[len(t) // 2], min(t)))
dev.close()
