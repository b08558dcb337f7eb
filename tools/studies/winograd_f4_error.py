#!/usr/bin/env python3
"""How much rounding error would F(4x4, 3x3) add to the fp32 trunk?  (A costing aid for DESIGN.md section 8: the fused F(4x4) kernel is
the one large fp32 lever left; its error decides whether it may serve the fp32 mode, whose kernels are tested at 1e-4 of the output
range and whose path is held to 1e-3.)  Emulates both algorithms in float32 exactly as a kernel would run them -- input and output
transforms in fp32, the transformed filter rounded once from float64, the channel contraction accumulated in fp32 -- on He-initialised
layers with post-ReLU inputs, against the float64 direct convolution.

    python tools/studies/winograd_f4_error.py [--cin 256] [--hw 48]
"""
import argparse

import numpy as np

# Lavin & Gray 2015, F(4x4, 3x3), interpolation points 0, +-1, +-2
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd(x, w, BT, G, AT, m):
    """x [Cin][H][W] float32 (H, W multiples of m), w [Cout][Cin][3][3] -> [Cout][H][W] float32, pad 1."""
    cin, H, W = x.shape
    cout = w.shape[0]
    a = m + 2
    xp = np.zeros((cin, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ij,ocjk,lk->iloc", G, w.astype(np.float64), G).astype(np.float32)          # rounded once
    BTf, ATf = BT.astype(np.float32), AT.astype(np.float32)
    out = np.zeros((cout, H, W), np.float32)
    for ty in range(H // m):
        for tx in range(W // m):
            d = xp[:, ty * m:ty * m + a, tx * m:tx * m + a]                                      # [cin][a][a]
            # fp32 transforms, one rounding per multiply-add as a kernel's v_fma chain would do
            t = np.einsum("ij,cjk->cik", BTf, d, dtype=np.float32, optimize=False)
            V = np.einsum("cik,lk->cil", t, BTf, dtype=np.float32, optimize=False)
            M = np.zeros((a, a, cout), np.float32)
            for c in range(cin):                                                                 # fp32 accumulation over channels
                M += U[:, :, :, c] * V[c][:, :, None]
            s = np.einsum("ij,jlo->ilo", ATf, M, dtype=np.float32, optimize=False)
            y = np.einsum("ilo,kl->iko", s, ATf, dtype=np.float32, optimize=False)
            out[:, ty * m:ty * m + m, tx * m:tx * m + m] = y.transpose(2, 0, 1)
    return out


def direct64(x, w):
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2), np.float64)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), np.float64)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, dy, dx].astype(np.float64), xp[:, dy:dy + H, dx:dx + W])
    return out


def direct32(x, w):
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), np.float32)
    for c in range(cin):
        for dy in range(3):
            for dx in range(3):
                out += w[:, c, dy, dx][:, None, None] * xp[c, dy:dy + H, dx:dx + W][None]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--hw", type=int, default=24)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    x = np.maximum(rng.normal(size=(args.cin, args.hw, args.hw)), 0).astype(np.float32)          # post-ReLU activations
    w = (rng.normal(size=(args.cout, args.cin, 3, 3)) * np.sqrt(2.0 / (9 * args.cin))).astype(np.float32)
    ref = direct64(x, w)
    rng_out = np.abs(ref).max()
    for name, y in (("direct fp32", direct32(x, w)), ("F(2x2,3x3) fp32", winograd(x, w, BT2, G2, AT2, 2)),
                    ("F(4x4,3x3) fp32", winograd(x, w, BT4, G4, AT4, 4))):
        e = np.abs(y - ref)
        print("%-16s Cin=%d  max |err| / output range = %.2e   rms err / rms out = %.2e" %
              (name, args.cin, e.max() / rng_out, np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())))


if __name__ == "__main__":
    main()
