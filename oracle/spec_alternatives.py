#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- blast radius of the SPEC-CHOICEs in oracle/SPEC.md.

ROIWarping / MaskResize / MaskPooling are specified by this project (their caffe-mnc source is not available: PARITY
UNPINNED).  For every convention that had to be chosen, this script evaluates the plausible ALTERNATIVE on BASELINE's fixture
(the seed-0 600x1000 image, seeded synthetic VGG-16 weights, 300 proposals) and reports how far the layer's own output and the
path's end outputs (21x21 masks, class probabilities, stage-2 boxes) move -- so that the day the caffe-mnc sources are mounted,
swapping a convention is a one-function change with a known effect.  The table it prints is pasted into oracle/SPEC.md section 6.

    python -m oracle.spec_alternatives            (CPU, about a minute)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import host, native  # noqa: E402
from oracle import net as onet  # noqa: E402


def roi_warp(feat, rois, PH, PW, scale, sample="topleft", round_edges=False, plus_one=True, oob="zero"):
    """SPEC.md section 1 with its choices as switches.  feat [1,C,H,W] or [C,H,W]; -> [R,C,PH,PW] float32."""
    f = np.asarray(feat, np.float32)
    f = f[0] if f.ndim == 4 else f
    C, H, W = f.shape
    one = np.float32(1.0)
    out = np.zeros((len(rois), C, PH, PW), np.float32)
    for r, roi in enumerate(np.asarray(rois, np.float32)):
        e = roi[1:5] * np.float32(scale)
        if round_edges:                                        # Fast R-CNN's ROIPooling rounds the scaled edges
            e = np.floor(e + np.float32(0.5)).astype(np.float32)
        x1s, y1s, x2s, y2s = e
        extra = one if plus_one else np.float32(0.0)
        rw, rh = max(x2s - x1s + extra, one), max(y2s - y1s + extra, one)
        bw, bh = np.float32(rw / PW), np.float32(rh / PH)
        px, py = np.arange(PW, dtype=np.float32), np.arange(PH, dtype=np.float32)
        if sample == "center":                                 # one sample at the centre of each bin
            sx, sy = x1s + (px + np.float32(0.5)) * bw, y1s + (py + np.float32(0.5)) * bh
        elif sample == "center_half":                          # bin centre in pixel-centre coordinates (RoIAlign's -0.5)
            sx, sy = x1s + (px + np.float32(0.5)) * bw - np.float32(0.5), y1s + (py + np.float32(0.5)) * bh - np.float32(0.5)
        else:                                                  # SPEC: top-left aligned, the paper's x0 + (u'/W') w
            sx, sy = x1s + px * bw, y1s + py * bh
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        ax, ay = (sx - x0).astype(np.float32), (sy - y0).astype(np.float32)

        def tap(yy, xx):
            if oob == "clamp":                                 # replicate the border instead of contributing 0
                return f[:, np.clip(yy, 0, H - 1)[:, None], np.clip(xx, 0, W - 1)[None, :]]
            v = f[:, np.clip(yy, 0, H - 1)[:, None], np.clip(xx, 0, W - 1)[None, :]]
            ok = ((yy >= 0) & (yy < H))[:, None] & ((xx >= 0) & (xx < W))[None, :]
            return v * ok[None].astype(np.float32)
        w00 = ((one - ax)[None, :] * (one - ay)[:, None])[None]
        w01 = (ax[None, :] * (one - ay)[:, None])[None]
        w10 = ((one - ax)[None, :] * ay[:, None])[None]
        w11 = (ax[None, :] * ay[:, None])[None]
        out[r] = w00 * tap(y0, x0) + w01 * tap(y0, x0 + 1) + w10 * tap(y0 + 1, x0) + w11 * tap(y0 + 1, x0 + 1)
    return out


def mask_resize(mask, OH, OW, mode="topleft"):
    """SPEC.md section 2 (mv_kernel.cu's top-left aligned resampling) or the two other common conventions."""
    m = np.asarray(mask, np.float32)
    R, _, IH, IW = m.shape

    def axis(O, I):
        d = np.arange(O, dtype=np.float32)
        if mode == "half_pixel":                               # cv2 / torch align_corners=False
            s = (d + np.float32(0.5)) * np.float32(I / O) - np.float32(0.5)
        elif mode == "align_corners":
            s = d * np.float32((I - 1) / (O - 1))
        else:
            s = d * np.float32(I / O)
        s = np.clip(s, 0, I - 1).astype(np.float32)
        lo = np.minimum(np.floor(s).astype(np.int64), I - 1)
        hi = np.minimum(lo + 1, I - 1)
        return lo, hi, (s - lo).astype(np.float32)
    y0, y1, fy = axis(OH, IH)
    x0, x1, fx = axis(OW, IW)
    one = np.float32(1.0)
    top = m[:, :, y0][:, :, :, x0] * (one - fx) + m[:, :, y0][:, :, :, x1] * fx
    bot = m[:, :, y1][:, :, :, x0] * (one - fx) + m[:, :, y1][:, :, :, x1] * fx
    return (top * (one - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


def mask_pool(feat, mask, binary=False):
    m = np.asarray(mask, np.float32)
    if binary:                                                 # CFM feeds binarised masks (TesterWrapper.py:398)
        m = (m >= np.float32(0.4)).astype(np.float32)
    return (np.asarray(feat, np.float32) * m).astype(np.float32)


ALTERNATIVES = [
    ("SPEC (numpy restatement vs oracle C)", {}, {}, {}),
    ("ROIWarping: sample at bin centres", {"sample": "center"}, {}, {}),
    ("ROIWarping: bin centres, pixel-centre coordinates (-0.5)", {"sample": "center_half"}, {}, {}),
    ("ROIWarping: scaled RoI edges rounded (as ROIPooling)", {"round_edges": True}, {}, {}),
    ("ROIWarping: width = x2 - x1 (no +1)", {"plus_one": False}, {}, {}),
    ("ROIWarping: border taps clamped instead of 0", {"oob": "clamp"}, {}, {}),
    ("MaskResize: half-pixel centres (cv2 / align_corners=False)", {}, {"mode": "half_pixel"}, {}),
    ("MaskResize: align_corners", {}, {"mode": "align_corners"}, {}),
    ("MaskPooling: mask binarised at 0.4 before the product", {}, {}, {"binary": True}),
]


def run_heads(w, c5, rois, im_info, warp_kw, resize_kw, pool_kw):
    """Both head stages with the given conventions patched into oracle.net (stage 2 on the alternative's own rois_ext)."""
    saved = (native.roi_warp, native.mask_resize, native.mask_pool)
    native.roi_warp = lambda feat, r, PH, PW, scale: roi_warp(feat, r, PH, PW, scale, **warp_kw)
    native.mask_resize = lambda m, OH, OW: mask_resize(m, OH, OW, **resize_kw)
    native.mask_pool = lambda f, m: mask_pool(f, m, **pool_kw)
    try:
        h1 = onet.head(w, c5, rois, False)
        rois_ext = host.stage_bridge_forward_test(rois, h1["bbox_pred"], h1["seg_cls_prob"], im_info)
        h2 = onet.head(w, c5, rois_ext, True)
    finally:
        native.roi_warp, native.mask_resize, native.mask_pool = saved
    return h1, rois_ext, h2


def main():
    import torch
    from mnc_amd import models, synth
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    proto = models.write_mnc_5stage_test_prototxt()
    w = synth.synthetic_weights(proto, seed=0)
    im = np.random.default_rng(0).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
    data, im_info, _ = host.prepare_mnc_args(im)
    c5 = onet.trunk(w, data)
    prob, bbox = onet.rpn(w, c5)
    rois = host.proposal_forward(prob, bbox, im_info)
    base1 = onet.head(w, c5, rois, False)
    base_ext = host.stage_bridge_forward_test(rois, base1["bbox_pred"], base1["seg_cls_prob"], im_info)
    base2 = onet.head(w, c5, base_ext, True)
    d = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
    dm = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).mean())
    rows = []
    for name, wk, rk, pk in ALTERNATIVES:
        h1, ext, h2 = run_heads(w, c5, rois, im_info, wk, rk, pk)
        layer = d(h1["roi_interpolate_conv5"], base1["roi_interpolate_conv5"]) if wk or not (rk or pk) else \
            d(h1["mask_proposal_resize"], base1["mask_proposal_resize"]) if rk else d(h1["roi_mask_conv5"], base1["roi_mask_conv5"])
        scale = float(np.abs(base1["roi_interpolate_conv5"]).max()) if (wk or not (rk or pk)) else 1.0 if rk else \
            float(np.abs(base1["roi_mask_conv5"]).max())
        flips = int((h1["seg_cls_prob"].argmax(1) != base1["seg_cls_prob"].argmax(1)).sum())
        rows.append((name, layer / scale, d(h1["mask_proposal"], base1["mask_proposal"]), dm(h1["mask_proposal"], base1["mask_proposal"]),
                     d(h1["seg_cls_prob"], base1["seg_cls_prob"]), dm(h1["seg_cls_prob"], base1["seg_cls_prob"]),
                     flips, d(ext, base_ext), dm(ext, base_ext), d(h2["mask_proposal"], base2["mask_proposal"]),
                     dm(h2["mask_proposal"], base2["mask_proposal"]), d(h2["seg_cls_prob"], base2["seg_cls_prob"])))
    print("| convention changed | layer output max (rel. to range) | stage-3 masks max / mean | stage-3 class prob. max / mean | "
          "arg-max class flips (of %d) | rois_ext px max / mean | stage-5 masks max / mean | stage-5 class prob. max |" % len(rois))
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %.1e | %.1e / %.1e | %.1e / %.1e | %d | %.1e / %.1e | %.1e / %.1e | %.1e |" % r)


if __name__ == "__main__":
    main()
