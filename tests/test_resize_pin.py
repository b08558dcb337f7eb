"""CPU: pins for the `cv2.resize(..., INTER_LINEAR)` restatement (lib/utils/blob.py:47-48 of the reference; oracle/SPEC.md section 5).

OpenCV is absent from this image and unpinned by the reference, so the oracle (oracle/host.py:resize_bilinear_cv) and the product's
numpy path (mnc_amd/lib/utils/blob.py:resize_linear; the GPU prep kernel is bit-identical to it) restate OpenCV's published
algorithm.  Until round 3 that restatement was checked only against itself.  Here it is held against two INDEPENDENT third-party
implementations of the same published convention (half-pixel centres, two taps per axis, no anti-aliasing):
  * torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False) on float32 -- ATen's kernel;
  * PIL.Image.resize(BILINEAR) on uint8 when enlarging (Pillow's filter support is one source pixel then, i.e. the same two taps),
    up to its 8-bit output rounding.
Sizes are the ones real inputs produce: VOC 375x500 -> 600x800 (scale 1.6), 427x640 -> 600x899, 600x512 -> 703x600, and a
reduction (CFM's 0.6 / 0.8 pyramid levels)."""
import numpy as np
import pytest

import mnc_amd
from oracle import host as ohost

mnc_amd.install_paths()

CASES = [((375, 500), 1.6), ((427, 640), 600.0 / 427.0), ((600, 512), 600.0 / 512.0), ((375, 500), 0.8), ((333, 500), 0.6)]


@pytest.mark.parametrize("hw,scale", CASES)
def test_resize_restatement_vs_torch(hw, scale):
    import torch
    import torch.nn.functional as F
    from utils.blob import resize_linear
    rng = np.random.default_rng(hw[0] + int(scale * 100))
    im = (rng.integers(0, 256, hw + (3,)).astype(np.float32) - np.array([102.9801, 115.9465, 122.7717], np.float32))
    want = ohost.resize_bilinear_cv(im, scale, scale)
    prod = resize_linear(im, scale, scale)
    assert np.array_equal(prod, want)                                    # product numpy path == oracle, bit for bit
    t = torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1)))[None]
    ref = F.interpolate(t, scale_factor=(scale, scale), mode="bilinear", align_corners=False, recompute_scale_factor=False)
    ref = ref[0].numpy().transpose(1, 2, 0)
    h, w = min(ref.shape[0], want.shape[0]), min(ref.shape[1], want.shape[1])      # floor vs round of the output size: <= 1 px
    assert abs(ref.shape[0] - want.shape[0]) <= 1 and abs(ref.shape[1] - want.shape[1]) <= 1
    # OpenCV (and the restatement) form the source coordinate in double and round it to float once, `(float)((dx+0.5)*scale - 0.5)`;
    # ATen forms it in float32.  At x ~ 640 that is a coordinate difference of up to 6e-5 px, times a gradient of up to 255 per px
    # (random pixels: the worst case) = 0.015.  A convention error (pixel-corner origin, align_corners) is O(10 - 100).
    d = np.abs(ref[:h, :w] - want[:h, :w])
    assert float(d.max()) < 0.05 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))


@pytest.mark.parametrize("hw,scale", CASES[:3])
def test_resize_restatement_vs_pillow_when_enlarging(hw, scale):
    from PIL import Image
    rng = np.random.default_rng(7)
    # smooth content (natural images are): Pillow rounds to 8 bits, the comparison is to one grey level
    base = rng.integers(0, 256, (hw[0] // 8 + 2, hw[1] // 8 + 2, 3)).astype(np.float32)
    im8 = np.clip(ohost.resize_bilinear_cv(base, 8.0, 8.0)[:hw[0], :hw[1]], 0, 255).astype(np.uint8)
    dh, dw = int(round(hw[0] * scale)), int(round(hw[1] * scale))
    # Pillow takes a size, i.e. factors in/out per axis: the dsize form of cv2.resize (resize_bilinear_cv_to; masks use it)
    want = ohost.resize_bilinear_cv_to(im8.astype(np.float32), dw, dh)
    pil = np.asarray(Image.fromarray(im8).resize((want.shape[1], want.shape[0]), Image.BILINEAR)).astype(np.float32)
    d = np.abs(pil - want)
    assert float(d.max()) <= 1.0 + 1e-3 and float(d.mean()) < 0.35, (float(d.max()), float(d.mean()))
