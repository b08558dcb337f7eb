"""GPU: the round-6 reduced-precision 3x3 convolution (csrc/conv_sw.hip, mnc_conv3x3_lowp) through the C ABI against torch-CPU
fp32 on the operands the mode rounds to (models/VGG16/mnc_5stage/test.prototxt:41-412 are its layers).

Tolerances: f16 / bf16 -- products of 2-byte values are exact in the fp32 accumulator, so against torch on the ROUNDED operands
only the summation order differs: 1e-5 of the output range; bf16x3 -- 1e-4 of the range against fp32 operands (the fp32 kernels'
bar; measured ~1e-5).  The packed output must be bit for bit mnc_act_pack of the fp32 output of the same launch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mnc_amd
from gpu_util import Dev, err, from_c8, to_c8
from mnc_amd import _lib

mnc_amd.install_paths()

pytestmark = pytest.mark.gpu

MODES = {"bf16x3": 0, "f16": 1, "bf16": 2}


@pytest.fixture(scope="module")
def dev():
    d = Dev(0)
    yield d
    d.close()


def _conv_ref(x, w, b, relu=True):
    y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), padding=1)
    return (F.relu(y) if relu else y)[0].numpy()


def _round(mode, a):
    if mode == "f16":
        return a.astype(np.float16).astype(np.float32)
    if mode == "bf16":
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
    return a


def _words(mode, n):
    return n if mode == "bf16x3" else n // 2


def run_lowp(dev, mode, x, w, b, relu, plan=None, want_packed=True, want_f32=True):
    """-> (fp32 output [Cout, H, W] or None, packed output words or None)"""
    m = MODES[mode]
    Cin, H, W = x.shape
    Cout = w.shape[0]
    nb = _lib.load().mnc_conv3x3_lowp_weight_bytes(m, Cout, Cin)
    assert nb > 0
    d_w = dev.empty((nb // 4,), fill=np.nan)
    dev.call("mnc_pack_conv3x3_lowp", m, dev.put(w), d_w, Cout, Cin)
    d_xp = dev.empty((_words(mode, Cin * H * W),), fill=np.nan)
    dev.call("mnc_act_pack", dev.put(to_c8(x)), d_xp, Cin * H * W, m)
    n_out = Cout * H * W
    d_y = dev.empty((n_out,), fill=np.nan) if want_f32 else None
    d_yp = dev.empty((_words(mode, n_out),), fill=np.nan) if want_packed else None
    if plan is not None:
        dev.tune("CONVX3_TILE", 100 + plan)
    try:
        dev.call("mnc_conv3x3_lowp", m, d_xp, d_w, dev.put(b), d_yp, d_y, H, W, Cin, Cout, relu)
    finally:
        if plan is not None:
            dev.tune("CONVX3_TILE", None)
    y = from_c8(dev.get(d_y, (n_out,)), Cout, H, W) if want_f32 else None
    yp = dev.get(d_yp, (_words(mode, n_out),)).view(np.uint32) if want_packed else None
    if want_f32 and want_packed:
        d_chk = dev.empty((_words(mode, n_out),), fill=np.nan)
        dev.call("mnc_act_pack", d_y, d_chk, n_out, m)
        chk = dev.get(d_chk, (_words(mode, n_out),)).view(np.uint32)
        assert np.array_equal(chk, yp), "packed output != pack(fp32 output)"
    return y, yp


def check(mode, got, x, w, b, relu):
    assert not np.isnan(got).any()
    if mode == "bf16x3":
        d, rel = err(got, _conv_ref(x, w, b, relu=bool(relu)))
        assert rel < 1e-4, (mode, d, rel)
    else:
        d, rel = err(got, _conv_ref(_round(mode, x), _round(mode, w), b, relu=bool(relu)))
        assert rel < 1e-5, (mode, d, rel)
    return rel


# (H, W, Cin, Cout): ragged edges, odd block counts (Cin % 16 == 8), every channel width, VGG-sized slices
SHAPES = [(6, 37, 16, 64), (9, 70, 8, 32), (38, 63, 64, 128), (13, 33, 128, 256), (75, 125, 32, 64), (4, 32, 8, 512),
          (80, 100, 24, 256), (150, 250, 16, 128), (11, 65, 40, 96), (1, 1, 8, 32), (2, 40, 16, 64), (5, 31, 72, 160)]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("H,W,Cin,Cout", SHAPES)
def test_conv3x3_lowp(dev, mode, H, W, Cin, Cout):
    rng = np.random.default_rng(H * 1000 + W + 17)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    for relu in (1, 0):
        y, _ = run_lowp(dev, mode, x, w, b, relu)
        rel = check(mode, y, x, w, b, relu)
    print("conv lowp %s %dx%d %d->%d: rel=%.3e" % (mode, H, W, Cin, Cout, rel))


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("plan", [0, 1, 2, 3])
def test_conv3x3_lowp_every_plan(dev, mode, plan):
    """Each (row groups, channel tiles, K ranges) instantiation at a small shape that fits it, same bits wanted from none of them
    (K ranges regroup the sums) but every one inside the mode's bar; one output form at a time as well."""
    H, W, Cin, Cout = 23, 70, 64, 128
    rng = np.random.default_rng(plan + 5)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    y, yp = run_lowp(dev, mode, x, w, b, 1, plan=plan)
    check(mode, y, x, w, b, 1)
    y2, _ = run_lowp(dev, mode, x, w, b, 1, plan=plan, want_packed=False)
    _, yp2 = run_lowp(dev, mode, x, w, b, 1, plan=plan, want_f32=False)
    assert np.array_equal(y, y2) and np.array_equal(yp, yp2)


@pytest.mark.parametrize("mode", list(MODES))
def test_conv3x3_lowp_vgg_layers(dev, mode):
    """The full-size layer classes of the 600x1000 trunk that pick plans 0, 1 and 2 on their own (conv3_1, conv4_1, conv5_1 at a
    quarter of their input channels: the oracle convolution stays in seconds)."""
    for H, W, Cin, Cout in ((150, 250, 32, 256), (75, 125, 64, 512), (38, 63, 128, 512)):
        rng = np.random.default_rng(H)
        x = np.maximum(rng.normal(size=(Cin, H, W)), 0).astype(np.float32)
        w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
        b = rng.normal(size=Cout).astype(np.float32)
        y, _ = run_lowp(dev, mode, x, w, b, 1)
        rel = check(mode, y, x, w, b, 1)
        print("conv lowp %s %dx%d %d->%d: rel=%.3e" % (mode, H, W, Cin, Cout, rel))


@pytest.mark.parametrize("mode", ["f16", "bf16x3"])
def test_conv3x3_lowp_plan_switch(dev, mode):
    """PLAN (round 6): the default plans are made for CU time -- the plain plan 0 down to 64 workgroups -- and PLAN=1 brings back the
    chip-filling choices of rounds 1-5: conv4_x (75x125, 512 channels) on plan 1, conv5_x (38x63) on plan 2.  Held bit for bit
    against the forced plan (CONVX3_TILE), at a quarter of the layers' input channels."""
    for H, W, Cin, Cout, lat_plan in ((75, 125, 64, 512, 1), (38, 63, 128, 512, 2)):
        rng = np.random.default_rng(H + 3)
        x = np.maximum(rng.normal(size=(Cin, H, W)), 0).astype(np.float32)
        w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
        b = rng.normal(size=Cout).astype(np.float32)
        forced = {p: run_lowp(dev, mode, x, w, b, 1, plan=p, want_packed=False)[0] for p in (0, lat_plan)}
        assert not np.array_equal(forced[0], forced[lat_plan])          # (the plans group the K sums differently)
        assert np.array_equal(run_lowp(dev, mode, x, w, b, 1, want_packed=False)[0], forced[0])
        dev.tune("PLAN", 1)
        try:
            assert np.array_equal(run_lowp(dev, mode, x, w, b, 1, want_packed=False)[0], forced[lat_plan])
        finally:
            dev.tune("PLAN", None)


def test_lowp_weight_layout(dev):
    """[chunk][Cout/32][plane][tap][32] x 8 values: f16 plane p = channels 8p..8p+7; bf16x3 planes (hi 0-7, hi 8-15, lo 0-7, lo 8-15)
    with hi + lo reproducing the weight to 2^-15; channels past Cin are zero."""
    rng = np.random.default_rng(0)
    Cout, Cin = 64, 24
    w = rng.normal(size=(Cout, Cin, 3, 3)).astype(np.float32)
    wp = np.zeros((Cout, 32, 3, 3), np.float32)
    wp[:, :Cin] = w
    lib = _lib.load()
    for mode, m in MODES.items():
        nb = lib.mnc_conv3x3_lowp_weight_bytes(m, Cout, Cin)
        npl = 4 if mode == "bf16x3" else 2
        assert nb == 2 * (Cout // 32) * npl * 9 * 32 * 16
        d_pk = dev.empty((nb // 4,), fill=np.nan)
        dev.call("mnc_pack_conv3x3_lowp", m, dev.put(w), d_pk, Cout, Cin)
        raw = dev.get(d_pk, (nb // 4,)).view(np.uint16).reshape(2, Cout // 32, npl, 9, 32, 8)
        if mode == "f16":
            val = raw.view(np.float16).astype(np.float32)
        else:
            val = (raw.astype(np.uint32) << 16).view(np.float32)
        # want[chunk, cot, plane(half), tap, col, e] = w[cot*32+col, chunk*16 + half*8 + e, tap]
        want = wp.reshape(Cout // 32, 32, 2, 2, 8, 9).transpose(2, 0, 3, 5, 1, 4)
        if mode == "bf16x3":
            rec = val[:, :, 0:2] + val[:, :, 2:4]
            nz = want != 0
            assert np.max(np.abs(rec - want)[nz] / np.abs(want)[nz]) < 2.0 ** -15
            assert not rec[~nz].any()
        else:
            assert np.array_equal(val, _round(mode, want))


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("H,W,Cin,Cout,plan", [(75, 125, 32, 64, None), (20, 64, 16, 128, 0), (23, 70, 64, 128, 1), (37, 33, 64, 128, 2),
                                               (10, 31, 16, 96, 3), (41, 97, 48, 128, 0), (30, 100, 32, 128, 1), (150, 250, 16, 256, None)])
def test_conv3x3_lowp_pool(dev, mode, H, W, Cin, Cout, plan):
    """The Pooling MAX 2x2/2 (ceil output size) folded into the epilogue == mnc_maxpool2_c8_<mode> of the unpooled packed output, bit
    for bit: odd heights and widths (a window of one row / one column), every plan (the one-row-group plans pool through the pooling kernel), K ranges inside the workgroup."""
    m = MODES[mode]
    rng = np.random.default_rng(H * 7 + W)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    nb = _lib.load().mnc_conv3x3_lowp_weight_bytes(m, Cout, Cin)
    d_w = dev.empty((nb // 4,), fill=np.nan)
    dev.call("mnc_pack_conv3x3_lowp", m, dev.put(w), d_w, Cout, Cin)
    d_xp = dev.empty((_words(mode, Cin * H * W),), fill=np.nan)
    dev.call("mnc_act_pack", dev.put(to_c8(x)), d_xp, Cin * H * W, m)
    d_b = dev.put(b)
    OH, OW = (H + 1) // 2, (W + 1) // 2
    for relu in (1, 0):
        d_full = dev.empty((_words(mode, Cout * H * W),), fill=np.nan)
        d_ref = dev.empty((_words(mode, Cout * OH * OW),), fill=np.nan)
        d_got = dev.empty((_words(mode, Cout * OH * OW),), fill=np.nan)
        if plan is not None:
            dev.tune("CONVX3_TILE", 100 + plan)
        try:
            dev.call("mnc_conv3x3_lowp", m, d_xp, d_w, d_b, d_full, None, H, W, Cin, Cout, relu)
            dev.call("mnc_conv3x3_lowp_pool", m, d_xp, d_w, d_b, d_got, H, W, Cin, Cout, relu)
        finally:
            if plan is not None:
                dev.tune("CONVX3_TILE", None)
        dev.call("mnc_maxpool2_c8_" + mode, d_full, d_ref, Cout, H, W)
        ref = dev.get(d_ref, (_words(mode, Cout * OH * OW),)).view(np.uint32)
        got = dev.get(d_got, (_words(mode, Cout * OH * OW),)).view(np.uint32)
        assert np.array_equal(ref, got), (mode, relu, int((ref != got).sum()))
