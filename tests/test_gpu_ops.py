"""GPU: every graph op of include/mnc_hip.h, called through the C ABI on device buffers, against the CPU oracle
(torch fp32 for the dense layers -- a floating-point kernel -- and oracle/mnc_oracle.c for the MNC-specific layers).

Tolerances (written per test): dense fp32 MFMA kernels accumulate in a different order than torch, so they are held to
1e-4 of the output's dynamic range (north_star allows 1e-3 end to end); gather/elementwise kernels mirror the oracle's
operation order and are held to 1e-6 relative / bit-exact where no FMA contraction can occur."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as GI
import mnc_amd
from gpu_util import Dev, err, from_c8, to_c8
from mnc_amd import _lib
from oracle import native

mnc_amd.install_paths()

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    d = Dev(0)
    yield d
    d.close()


TUNING_BUILD = b"tuning" in _lib.load().mnc_version()     # built with MNC_HIPCC_EXTRA=-DMNC_TUNING (ablation / superseded kernels)


@pytest.fixture
def tune(dev):
    """tune(name, value): override one of the launchers' choices on the module's context (mnc_ctx_set_tuning) for this test."""
    keys = []

    def set_(name, value):
        keys.append(name)
        dev.tune(name, value)
    yield set_
    for k in keys:
        dev.tune(k, None)


def _conv_ref(x, w, b, relu=True, pad=1):
    y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), padding=pad)
    return (F.relu(y) if relu else y)[0].numpy()


# (H, W, Cin, Cout): ragged edges, every channel-tile width the dispatcher can pick, and a VGG-sized slice
def _lowp_w(dev, mode, Cout, Cin):
    """Device buffer for mnc_pack_conv3x3_{bf16x3,f16,bf16}: mnc_conv3x3_lowp_weight_bytes of the mode."""
    nb = _lib.load().mnc_conv3x3_lowp_weight_bytes({"bf16x3": 0, "f16": 1, "bf16": 2}[mode], Cout, Cin)
    assert nb > 0 and nb % 4 == 0
    return dev.empty((nb // 4,), fill=np.nan)


CONV_SHAPES = [(6, 37, 16, 64), (9, 70, 8, 32), (38, 63, 64, 128), (13, 33, 128, 256), (75, 125, 32, 64), (4, 32, 8, 512)]


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES)
@pytest.mark.parametrize("relu", [1, 0])
def test_conv3x3_mfma(dev, H, W, Cin, Cout, relu):
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    d_w_raw, d_b = dev.put(w), dev.put(b)
    d_w = dev.empty(((Cin // 8) * Cout * 76,))
    dev.call("mnc_pack_conv3x3_weights", d_w_raw, d_w, Cout, Cin)
    d_x = dev.put(to_c8(x))
    d_y = dev.empty((Cout * H * W,), fill=np.nan)
    dev.call("mnc_conv3x3", d_x, d_w, d_b, d_y, H, W, Cin, Cout, relu)
    got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
    want = _conv_ref(x, w, b, relu=bool(relu))
    assert not np.isnan(got).any()
    d, rel = err(got, want)
    assert rel < 1e-4, (d, rel)


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES + [(150, 250, 16, 128), (75, 125, 64, 64), (5, 3, 8, 32), (38, 63, 128, 64),
                                                        (75, 125, 256, 512), (37, 63, 256, 512)])
@pytest.mark.parametrize("relu", [1, 0])
def test_conv3x3_winograd(dev, monkeypatch, H, W, Cin, Cout, relu, tune):
    """mnc_conv3x3_wino (Winograd F(2x2,3x3) on the fp32 matrix pipe) against torch fp32 and against the direct kernel: odd
    heights / widths (partial 2x2 tiles at the border), every workgroup height, K splits.  The transforms are exact in fp32
    except for the summation order, so the bar is the direct kernel's own (1e-4 of the output range; measured ~1e-6).
    The default run of the last two shapes takes the launcher's tail plan: (75,125,256,512) 512 unsplit workgroups + 128 tiles
    in 4 K ranges, (37,63,256,512) every tile in 3 uneven ranges (32 blocks) with odd H and W under the pooling reduce."""
    rng = np.random.default_rng(H * 1000 + W + Cin + Cout)
    x = rng.normal(0, 1, (Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(0, 0.1, Cout).astype(np.float32)
    want = _conv_ref(x, w, b, bool(relu))
    d_x, d_b = dev.put(to_c8(x)), dev.put(b)
    d_raw = dev.put(w)
    d_w = dev.empty((Cin * Cout * 17,))
    dev.call("mnc_pack_conv3x3_wino", d_raw, d_w, Cout, Cin)
    d_wd = dev.empty(((Cin // 8) * Cout * 76,))
    dev.call("mnc_pack_conv3x3_weights", d_raw, d_wd, Cout, Cin)
    d_y = dev.empty((Cout, H, W), fill=-7.0)
    dev.call("mnc_conv3x3", d_x, d_wd, d_b, d_y, H, W, Cin, Cout, relu)
    direct = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
    # (rows, K splits, kernel build): the default plan; forced workgroup heights / uniform K splits; the older loop builds
    # (MNC_WINO_VAR 3: rotated loop, 1: flat block with register staging, 0: round-2 v2 loop) and the default without the tail plan
    for rows, ks, var in ((None, None, None), ("1", "1", None), ("2", "2", None), ("4", "1", None), ("1", "4", None),
                          (None, None, "3"), (None, None, "1"), (None, None, "0"), ("2", "2", "1"), (None, None, "notail"),
                          (None, None, "stream1"), (None, None, "stream2"), (None, None, "stream5"), (None, None, "mfma16_0"),
                          (None, None, "mfma16_1")):
        if ks is not None and (Cin // 8) % int(ks):
            continue
        if (var in ("3", "1", "0") or (var or "").startswith(("stream", "mfma16"))) and not TUNING_BUILD:   # superseded / measurement builds: -DMNC_TUNING only
            continue
        for k in ("WINO_ROWS", "CONV_KSPLIT", "WINO_VAR", "WINO_TAIL", "WINO_STREAM", "WINO_MFMA16"):
            dev.tune(k, None)
        if rows is not None:
            tune("WINO_ROWS", rows)
            tune("CONV_KSPLIT", ks)
        if var == "notail":
            tune("WINO_TAIL", "0")
        elif var is not None and var.startswith("mfma16"):       # the 16 x 16 x 4 build (conv_wino16.hip) on / off
            tune("WINO_MFMA16", var[7:])
        elif var is not None and var.startswith("stream"):       # the layer as one stream of units, 512 k ranges (conv_wino_stream.hip)
            tune("WINO_STREAM", var[6:])
        elif var is not None:
            tune("WINO_VAR", var)
        dev.put_into(d_y, np.full((Cout, H, W), -7.0, np.float32))
        dev.call("mnc_conv3x3_wino", d_x, d_w, d_b, d_y, H, W, Cin, Cout, relu)
        got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
        assert err(got, want)[1] < 1e-4, (rows, ks, var, err(got, want))
        assert err(got, direct)[1] < 1e-5, (rows, ks, var, err(got, direct))
        # + the following Pooling MAX 2x2/2 in the epilogue (Caffe's ceil output size: odd H / W clip the last window), with and
        # without K splits: exactly the maximum over the un-fused kernel's own outputs
        if rows in (None, "2") and H >= 2 and W >= 2:     # (every build above fuses the pooling)
            OH, OW = (H + 1) // 2, (W + 1) // 2
            d_p = dev.empty((Cout, OH, OW), fill=-7.0)
            dev.call("mnc_conv3x3_wino_pool", d_x, d_w, d_b, d_p, H, W, Cin, Cout, relu)
            pooled = from_c8(dev.get(d_p, (Cout * OH * OW,)), Cout, OH, OW)
            ref = F.max_pool2d(torch.from_numpy(got)[None], 2, 2, ceil_mode=True)[0].numpy()
            assert np.array_equal(pooled, ref), (rows, ks, var)


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES + [(150, 250, 16, 128), (5, 3, 8, 32), (38, 63, 128, 64), (75, 125, 256, 512),
                                                        (37, 63, 256, 512), (150, 250, 64, 256), (300, 500, 64, 128)])
@pytest.mark.parametrize("relu", [1, 0])
def test_conv3x3_winograd_f4(dev, H, W, Cin, Cout, relu, tune):
    """mnc_conv3x3_wino4 (Winograd F(4x4,3x3) fused on the fp32 matrix pipe, csrc/conv_wino4.hip) against torch fp32: odd heights /
    widths (partial 4x4 tiles, partial 16x64 workgroup tiles, maps smaller than one tile), one and several channel tiles, the
    launcher's plans -- unsplit, uniform K ranges (maps that do not fill the chip: (75,125,256,512) three ranges, (37,63,256,512)
    five uneven ones), the tail plan ((150,250,64,256): 32 whole tiles + 8 tiles in two ranges; (300,500,64,128)) -- and forced
    uniform cuts; both block orders.  The bar is the fp32 kernels' 1e-4 of the output range (F(4x4)'s transforms multiply by up to
    8 and 1/24: measured ~1e-5 at 256 input channels, printed).  With the fused Pooling the result is exactly the maximum over the
    un-fused kernel's own outputs under the same plan."""
    rng = np.random.default_rng(H * 1000 + W + Cin + Cout)
    x = rng.normal(0, 1, (Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(0, 0.1, Cout).astype(np.float32)
    want = _conv_ref(x, w, b, bool(relu))
    d_x, d_b = dev.put(to_c8(x)), dev.put(b)
    d_w = dev.empty((Cin * Cout * 36,), fill=np.nan)
    dev.call("mnc_pack_conv3x3_wino4", dev.put(w), d_w, Cout, Cin)
    assert not np.isnan(dev.get(d_w, (Cin * Cout * 36,))).any()
    d_y = dev.empty((Cout, H, W), fill=-7.0)
    worst = 0.0
    for ks, tail, xcd in ((None, None, None), ("1", None, None), ("2", None, "0"), ("3", None, "1"), (None, "0", None), (None, "1", None)):
        if ks is not None and int(ks) > Cin // 8:
            continue
        for k in ("CONV_KSPLIT", "WINO_TAIL", "WINO_XCD"):
            dev.tune(k, None)
        if ks is not None:
            tune("CONV_KSPLIT", ks)
        if tail is not None:
            tune("WINO_TAIL", tail)
        if xcd is not None:
            tune("WINO_XCD", xcd)
        dev.put_into(d_y, np.full((Cout, H, W), -7.0, np.float32))
        dev.call("mnc_conv3x3_wino4", d_x, d_w, d_b, d_y, H, W, Cin, Cout, relu)
        got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
        rel = err(got, want)[1]
        worst = max(worst, rel)
        assert rel < 1e-4, (ks, tail, xcd, err(got, want))
        # K ranges summed inside the launch by each tile's last arriver (default) == summed by wino4_section_reduce_kernel
        # (FC_REDUCE=0): the same additions in the same order, bit for bit, whichever workgroup arrives last
        tune("FC_REDUCE", "0")
        dev.put_into(d_y, np.full((Cout, H, W), -7.0, np.float32))
        dev.call("mnc_conv3x3_wino4", d_x, d_w, d_b, d_y, H, W, Cin, Cout, relu)
        assert np.array_equal(got, from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)), (ks, tail, xcd)
        dev.tune("FC_REDUCE", None)
        if H >= 2 and W >= 2:
            OH, OW = (H + 1) // 2, (W + 1) // 2
            d_p = dev.empty((Cout, OH, OW), fill=-7.0)
            dev.call("mnc_conv3x3_wino4_pool", d_x, d_w, d_b, d_p, H, W, Cin, Cout, relu)
            pooled = from_c8(dev.get(d_p, (Cout * OH * OW,)), Cout, OH, OW)
            ref = F.max_pool2d(torch.from_numpy(got)[None], 2, 2, ceil_mode=True)[0].numpy()
            assert np.array_equal(pooled, ref), (ks, tail, xcd)
    print("conv3x3 F(4x4) %dx%d %d->%d relu=%d: max rel err %.2e" % (H, W, Cin, Cout, relu, worst))


@pytest.mark.parametrize("H,W,Cin,Cout", [(150, 250, 64, 256), (75, 125, 256, 512), (300, 500, 16, 64), (19, 25, 32, 32), (10, 13, 64, 64),
                                          (5, 7, 64, 64), (38, 50, 16, 32)])
def test_conv3x3_winograd_f4_is_bit_reproducible(dev, H, W, Cin, Cout):
    """The same launch forty (small maps: a hundred) times: every result bit-identical to the first.  The F(4x4) kernel refills its
    LDS buffers by DMA while other waves of the workgroup read their neighbours; a missing barrier shows up as a difference between
    runs, not necessarily as an error against the reference.  A weak guard on its own: round 4's race (a fast wave refilled halo
    buffer 0 while a slow one was still reading it in its prologue) passed this test and every unit test -- back-to-back launches
    of one kernel keep the waves in step -- and failed tests/test_gpu_pipeline.py::test_image_stream_returns_results_in_order six
    times out of six, where other streams' kernels share the CUs.  That test is the guard; this one covers the shapes it runs."""
    rng = np.random.default_rng(H + W + Cin + Cout)
    x = rng.normal(0, 1, (Cin, H, W)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(0, 0.1, Cout).astype(np.float32)
    d_x, d_b = dev.put(to_c8(x)), dev.put(b)
    d_w = dev.empty((Cin * Cout * 36,))
    dev.call("mnc_pack_conv3x3_wino4", dev.put(w), d_w, Cout, Cin)
    d_y = dev.empty((Cout, H, W), fill=-7.0)
    first = None
    for rep in range(40 if H * W > 2000 else 100):
        dev.call("mnc_conv3x3_wino4", d_x, d_w, d_b, d_y, H, W, Cin, Cout, 1)
        got = dev.get(d_y, (Cout * H * W,)).copy()
        if first is None:
            first = got
        else:
            assert np.array_equal(first, got), "run %d differs from run 0 in %d values" % (rep, int((first != got).sum()))


@pytest.mark.parametrize("H,W,Cin,Cout", [(75, 125, 256, 512), (37, 63, 256, 512), (150, 250, 64, 256)])
def test_conv3x3_winograd_f4_in_launch_reduction_reads_fresh_slabs(dev, tune, H, W, Cin, Cout):
    """The K ranges of a launch are summed by each tile's last arriver from slabs the other workgroups published (mnc_internal.h).
    Repeating ONE launch cannot see a stale slab (the stale bytes equal the fresh ones), so two different inputs alternate through
    the same slab addresses: every result must equal that input's own result under the separate reduction kernel."""
    rng = np.random.default_rng(H + W + Cin + 5)
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(0, 0.1, Cout).astype(np.float32)
    d_b = dev.put(b)
    d_w = dev.empty((Cin * Cout * 36,))
    dev.call("mnc_pack_conv3x3_wino4", dev.put(w), d_w, Cout, Cin)
    d_y = dev.empty((Cout, H, W), fill=-7.0)
    xs = [dev.put(to_c8(rng.normal(0, 1 + i, (Cin, H, W)).astype(np.float32))) for i in range(2)]
    tune("FC_REDUCE", "0")
    refs = []
    for d_x in xs:
        dev.call("mnc_conv3x3_wino4", d_x, d_w, d_b, d_y, H, W, Cin, Cout, 1)
        refs.append(dev.get(d_y, (Cout * H * W,)).copy())
    dev.tune("FC_REDUCE", None)
    assert not np.array_equal(refs[0], refs[1])
    for rep in range(12):
        k = rep & 1
        dev.call("mnc_conv3x3_wino4", xs[k], d_w, d_b, d_y, H, W, Cin, Cout, 1)
        got = dev.get(d_y, (Cout * H * W,))
        assert np.array_equal(got, refs[k]), "launch %d (input %d): %d values differ" % (rep, k, int((got != refs[k]).sum()))


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES + [(150, 250, 16, 128), (80, 100, 24, 256)])
@pytest.mark.parametrize("relu", [1, 0])
def test_conv3x3_bf16x3(dev, H, W, Cin, Cout, relu):
    """Split-precision conv (3 bf16 MFMAs per product) against torch fp32; same 1e-4-of-range bar as the fp32 kernel
    (measured error ~1e-5); fp32 tensors in and out (packed into the scratch arena first).  Round 6: csrc/conv_sw.hip."""
    rng = np.random.default_rng(H * 1000 + W + 7)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    d_w = _lowp_w(dev, "bf16x3", Cout, Cin)
    dev.call("mnc_pack_conv3x3_bf16x3", dev.put(w), d_w, Cout, Cin)
    d_y = dev.empty((Cout * H * W,), fill=np.nan)
    dev.call("mnc_conv3x3_bf16x3", dev.put(to_c8(x)), d_w, dev.put(b), d_y, H, W, Cin, Cout, relu)
    got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
    want = _conv_ref(x, w, b, relu=bool(relu))
    assert not np.isnan(got).any()
    d, rel = err(got, want)
    print("conv bf16x3 %dx%d %d->%d: max|d|=%.3e rel=%.3e" % (H, W, Cin, Cout, d, rel))
    assert rel < 1e-4, (d, rel)


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES + [(150, 250, 16, 128)])
def test_conv3x3_f16(dev, H, W, Cin, Cout):
    """f16 math mode: exact against torch on fp16-rounded operands (fp32 accumulation), ~3e-4 of range against fp32."""
    rng = np.random.default_rng(H * 1000 + W + 9)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    d_w = _lowp_w(dev, "f16", Cout, Cin)
    dev.call("mnc_pack_conv3x3_f16", dev.put(w), d_w, Cout, Cin)
    d_y = dev.empty((Cout * H * W,), fill=np.nan)
    dev.call("mnc_conv3x3_f16", dev.put(to_c8(x)), d_w, dev.put(b), d_y, H, W, Cin, Cout, 1)
    got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    assert not np.isnan(got).any()
    d, rel = err(got, _conv_ref(r16(x), r16(w), b, relu=True))
    d32, rel32 = err(got, _conv_ref(x, w, b, relu=True))
    print("conv f16 %dx%d %d->%d: vs fp16-rounded operands rel=%.3e, vs fp32 rel=%.3e" % (H, W, Cin, Cout, rel, rel32))
    assert rel < 1e-5 and rel32 < 2e-3


def _rbf16(a):
    """float32 -> nearest-even bf16 -> float32 (what v_cvt_pk_bf16_f32 does)"""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("H,W,Cin,Cout", CONV_SHAPES + [(150, 250, 16, 128)])
def test_conv3x3_bf16(dev, H, W, Cin, Cout):
    """Plain "bf16" math mode (BASELINE configs[2] as written, round 4): exact against torch on bf16-rounded operands (products of
    bf16 values are exact in the fp32 accumulator), ~4e-3 of range against fp32 -- the figure that keeps this mode out of the 1e-3 bar."""
    rng = np.random.default_rng(H * 1000 + W + 13)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    d_w = _lowp_w(dev, "bf16", Cout, Cin)
    dev.call("mnc_pack_conv3x3_bf16", dev.put(w), d_w, Cout, Cin)
    d_y = dev.empty((Cout * H * W,), fill=np.nan)
    dev.call("mnc_conv3x3_bf16", dev.put(to_c8(x)), d_w, dev.put(b), d_y, H, W, Cin, Cout, 1)
    got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
    assert not np.isnan(got).any()
    d, rel = err(got, _conv_ref(_rbf16(x), _rbf16(w), b, relu=True))
    d32, rel32 = err(got, _conv_ref(x, w, b, relu=True))
    print("conv bf16 %dx%d %d->%d: vs bf16-rounded operands rel=%.3e, vs fp32 rel=%.3e" % (H, W, Cin, Cout, rel, rel32))
    assert rel < 1e-5 and rel32 < 2e-2


@pytest.mark.parametrize("M,N,K,act", [(300, 256, 1024, 1), (300, 4096, 4096, 1), (37, 130, 512, 0), (300, 441, 256, 2),
                                       (1000, 4096, 12544, 1), (300, 4096, 25088, 1)])
def test_fc_bf16(dev, M, N, K, act):
    """mnc_fc_bf16: exact against torch on operands rounded to bf16, within bf16's 8 bits of the fp32 product; small, 320-row and
    256-column LDS-DMA kernel shapes."""
    rng = np.random.default_rng(M + N + K + 1)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_wp = dev.empty(((N + 127) // 128 * 128 * K // 2,), fill=np.nan)
    dev.call("mnc_pack_fc_bf16", dev.put(w), d_wp, N, K)
    d_o = dev.empty((M * N,), fill=np.nan)
    dev.call("mnc_fc_bf16", dev.put(a), d_wp, dev.put(b), d_o, M, N, K, N, act)
    got = dev.get(d_o, (M, N))

    def ref(x, y):
        z = F.linear(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(b))
        return (F.relu(z) if act == 1 else torch.sigmoid(z) if act == 2 else z).numpy()
    assert not np.isnan(got).any()
    d, rel = err(got, ref(_rbf16(a), _rbf16(w)))
    d32, rel32 = err(got, ref(a, w))
    print("bf16 M=%d N=%d K=%d: vs bf16-rounded operands rel=%.3e, vs fp32 rel=%.3e" % (M, N, K, rel, rel32))
    assert rel < 1e-5 and rel32 < 2e-2


@pytest.mark.parametrize("mode", ["bf16x3", "f16", "bf16"])
@pytest.mark.parametrize("H,W,Cin,Cout", [(9, 70, 8, 32), (38, 63, 64, 128), (13, 33, 128, 256), (75, 125, 32, 64), (150, 250, 16, 128)])
def test_conv3x3_packed_activations(dev, mode, H, W, Cin, Cout):
    """2-byte activations between MFMA layers: a producer that writes the packed form and a consumer that reads it give bit for
    bit what the fp32-tensor route gives (the producer's epilogue applies the consumer's own split / rounding); the packed MAX
    2x2/2 pool equals packing the pooled fp32 tensor.  Shapes cover every plan of conv_sw.hip's launcher."""
    f16 = {"bf16x3": 0, "f16": 1, "bf16": 2}[mode]          # mnc_act_pack's format number
    rng = np.random.default_rng(H * 1000 + W + 11)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    d_w = _lowp_w(dev, mode, Cout, Cin)
    dev.call("mnc_pack_conv3x3_" + mode, dev.put(w), d_w, Cout, Cin)
    d_b, d_x = dev.put(b), dev.put(to_c8(x))
    n_in, n_out = Cin * H * W, Cout * H * W
    words = lambda n: n // 2 if f16 else n                      # 32-bit words of a packed tensor of n elements
    d_xp = dev.empty((words(n_in),), fill=np.nan)
    dev.call("mnc_act_pack", d_x, d_xp, n_in, f16)
    # reference route: fp32 tensors in and out
    d_y = dev.empty((n_out,), fill=np.nan)
    dev.call("mnc_conv3x3_" + mode, d_x, d_w, d_b, d_y, H, W, Cin, Cout, 1)
    y = dev.get(d_y, (n_out,))
    d_yp = dev.empty((words(n_out),), fill=np.nan)
    dev.call("mnc_act_pack", d_y, d_yp, n_out, f16)
    yp = dev.get(d_yp, (words(n_out),)).view(np.uint32)
    pk = "mnc_conv3x3_%s_pk" % mode
    for in_pk, out_pk in ((1, 0), (0, 1), (1, 1)):
        d_o = dev.empty((words(n_out) if out_pk else n_out,), fill=np.nan)
        dev.call(pk, d_xp if in_pk else d_x, d_w, d_b, d_o, H, W, Cin, Cout, 1, in_pk, out_pk)
        got = dev.get(d_o, (words(n_out) if out_pk else n_out,)).view(np.uint32)
        assert np.array_equal(got, yp if out_pk else y.view(np.uint32)), (in_pk, out_pk)
    # unpack(pack(y)) is y to the format's precision
    d_u = dev.empty((n_out,), fill=np.nan)
    dev.call("mnc_act_unpack", d_yp, d_u, n_out, f16)
    u = dev.get(d_u, (n_out,))
    assert np.abs(u - y).max() <= ((1e-3 if f16 == 1 else 2.0 ** -8) if f16 else 2.0 ** -15) * np.abs(y).max()
    if f16 == 1:
        assert np.array_equal(u, y.astype(np.float16).astype(np.float32))
    elif f16 == 2:
        assert np.array_equal(u, _rbf16(y))
    # pool: packed(pool(y)) == pool_packed(packed(y))
    OH, OW = (H + 1) // 2, (W + 1) // 2
    d_p = dev.empty((Cout * OH * OW,), fill=np.nan)
    dev.call("mnc_maxpool2_c8", d_y, d_p, Cout, H, W)
    d_pp = dev.empty((words(Cout * OH * OW),), fill=np.nan)
    dev.call("mnc_act_pack", d_p, d_pp, Cout * OH * OW, f16)
    d_q = dev.empty((words(Cout * OH * OW),), fill=np.nan)
    dev.call("mnc_maxpool2_c8_" + mode, d_yp, d_q, Cout, H, W)
    a, c = dev.get(d_pp, (words(Cout * OH * OW),)).view(np.uint32), dev.get(d_q, (words(Cout * OH * OW),)).view(np.uint32)
    assert np.array_equal(a, c)


def test_conv3x3_packed_weight_layout(dev):
    rng = np.random.default_rng(0)
    Cout, Cin = 64, 16
    w = rng.normal(size=(Cout, Cin, 3, 3)).astype(np.float32)
    d_raw = dev.put(w)
    d_pk = dev.empty(((Cin // 8) * Cout * 76,))
    dev.call("mnc_pack_conv3x3_weights", d_raw, d_pk, Cout, Cin)
    pk = dev.get(d_pk, (Cin // 8, Cout, 76))
    assert not pk[:, :, 72:].any()
    for cb in range(Cin // 8):
        for tap in range(9):
            assert np.array_equal(pk[cb, :, tap * 8:tap * 8 + 8], w[:, cb * 8:cb * 8 + 8, tap // 3, tap % 3])


@pytest.mark.parametrize("H,W", [(20, 33), (600, 1000)])
def test_conv1_1_direct(dev, H, W, tune):
    """conv1_1 (3 input channels, NCHW in, c8 out) against torch: the matrix-pipe kernel (Cout a multiple of 16 up to 64: im2col on
    the fly, 16 pixels x 64 channels per wave step) and the VALU kernel (other widths; CONV_COT=-1 forces it) -- every width the
    launcher splits on, ragged pixel tiles, no ReLU."""
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, H, W)).astype(np.float32)
    for Cout, relu in ((64, 1), (16, 0), (48, 1), (8, 1), (72, 0)):
        w = (rng.normal(size=(Cout, 3, 3, 3)) * 0.2).astype(np.float32)
        b = rng.normal(size=Cout).astype(np.float32)
        want = _conv_ref(x, w, b, relu=bool(relu))
        d_x, d_w, d_b = dev.put(x), dev.put(w), dev.put(b)
        d_y = dev.empty((Cout * H * W,), fill=np.nan)
        dev.call("mnc_conv3x3_c3", d_x, d_w, d_b, d_y, H, W, Cout, relu)
        got = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
        assert not np.isnan(got).any()
        _, rel = err(got, want)
        assert rel < 1e-5, (Cout, rel)
        tune("CONV_COT", "-1")
        dev.put_into(d_y, np.full((Cout * H * W,), np.nan, np.float32))
        dev.call("mnc_conv3x3_c3", d_x, d_w, d_b, d_y, H, W, Cout, relu)
        valu = from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)
        dev.tune("CONV_COT", None)
        assert err(valu, want)[1] < 1e-5 and err(valu, got)[1] < 2e-6, Cout


@pytest.mark.parametrize("C,H,W", [(16, 75, 125), (8, 2, 2), (64, 600, 1000), (32, 37, 64)])
def test_maxpool2_c8_ceil_mode(dev, C, H, W):
    x = np.random.default_rng(2).normal(size=(C, H, W)).astype(np.float32)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    d_y = dev.empty((C * OH * OW,), fill=np.nan)
    dev.call("mnc_maxpool2_c8", dev.put(to_c8(x)), d_y, C, H, W)
    got = from_c8(dev.get(d_y, (C * OH * OW,)), C, OH, OW)
    want = F.max_pool2d(torch.from_numpy(x)[None], 2, 2, ceil_mode=True)[0].numpy()
    assert want.shape == (C, OH, OW) and np.array_equal(got, want)
    assert np.array_equal(got, native.maxpool2(x))


def test_conv1x1_and_rpn_softmax(dev):
    rng = np.random.default_rng(3)
    H, W, Cin = 38, 63, 512
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(18, Cin)) * 0.05).astype(np.float32)
    b = rng.normal(size=18).astype(np.float32)
    d_s = dev.empty((18 * H * W,), fill=np.nan)
    dev.call("mnc_conv1x1_to_nchw", dev.put(to_c8(x)), dev.put(w), dev.put(b), d_s, H, W, Cin, 18)
    score = dev.get(d_s, (18, H, W))
    want = _conv_ref(x, w.reshape(18, Cin, 1, 1), b, relu=False, pad=0)
    _, rel = err(score, want)
    assert rel < 1e-5, rel
    d_p = dev.empty((18 * H * W,), fill=np.nan)
    dev.call("mnc_rpn_softmax", d_s, d_p, 9, H, W)
    prob = dev.get(d_p, (18, H, W))
    t = torch.from_numpy(score)[None]
    wantp = F.softmax(t.reshape(1, 2, -1, W), dim=1).reshape(1, 18, H, W)[0].numpy()
    assert err(prob, wantp)[0] < 1e-6
    assert np.abs(prob[:9] + prob[9:] - 1).max() < 1e-6
    # both heads and the softmax as ONE launch (mnc_rpn_heads: what the whole-image pipeline runs): the bits of the three calls;
    # a reduced-width RPN (Cin % 16 != 0) and a ragged last pixel tile too
    for (H2, W2, C2, A) in ((38, 63, 512, 9), (7, 9, 24, 9), (5, 13, 64, 3), (6, 8, 32, 16)):
        x2 = rng.normal(size=(C2, H2, W2)).astype(np.float32)
        w2 = (rng.normal(size=(6 * A, C2)) * 0.05).astype(np.float32)
        b2 = rng.normal(size=6 * A).astype(np.float32)
        d_x2, d_w2, d_b2 = dev.put(to_c8(x2)), dev.put(w2), dev.put(b2)
        d_s2, d_p2 = dev.empty((6 * A * H2 * W2,), fill=np.nan), dev.empty((2 * A * H2 * W2,), fill=np.nan)
        dev.call("mnc_conv1x1_to_nchw", d_x2, d_w2, d_b2, d_s2, H2, W2, C2, 6 * A)
        dev.call("mnc_rpn_softmax", d_s2, d_p2, A, H2, W2)
        s_sep, p_sep = dev.get(d_s2, (6 * A, H2, W2)).copy(), dev.get(d_p2, (2 * A, H2, W2)).copy()
        ref = _conv_ref(x2, w2.reshape(6 * A, C2, 1, 1), b2, relu=False, pad=0)
        assert err(s_sep, ref)[1] < 1e-5
        dev.put_into(d_s2, np.full((6 * A, H2, W2), np.nan, np.float32))
        dev.put_into(d_p2, np.full((2 * A, H2, W2), np.nan, np.float32))
        dev.call("mnc_rpn_heads", d_x2, d_w2, d_b2, d_s2, d_p2, H2, W2, C2, A)
        assert np.array_equal(dev.get(d_s2, (6 * A, H2, W2)), s_sep), (H2, W2, C2, A)
        assert np.array_equal(dev.get(d_p2, (2 * A, H2, W2)), p_sep), (H2, W2, C2, A)


GEN_CONV = [  # H, W, Cin, Cout, K, stride, pad, residual   (ResNet-50 shapes in miniature, plus ragged tiles)
    (37, 53, 64, 256, 1, 1, 0, False), (37, 53, 256, 64, 1, 1, 0, True), (40, 54, 256, 128, 1, 2, 0, False),
    (33, 47, 64, 64, 3, 2, 1, False), (21, 30, 128, 72, 3, 1, 1, True), (12, 9, 8, 8, 5, 1, 2, False),
    (50, 84, 1024, 256, 1, 1, 0, False),
    (200, 334, 64, 256, 1, 1, 0, True), (150, 250, 24, 136, 3, 2, 1, False)]     # large grids


@pytest.mark.parametrize("H,W,Cin,Cout,K,stride,pad,residual", GEN_CONV)
@pytest.mark.parametrize("relu,wide", [(1, False), (0, False), (1, True)])
def test_conv2d_general(dev, monkeypatch, H, W, Cin, Cout, K, stride, pad, residual, relu, wide, tune):
    """mnc_conv2d (any kernel / stride / pad, + bias + residual + ReLU) against torch fp32; `wide` = the 128-channel workgroup
    tile (MNC_CONV2D_WIDE, a tuning knob -- the 64-channel tile is the default)."""
    if wide:
        tune("CONV2D_WIDE", "1")
    rng = np.random.default_rng(H * 100 + W + K)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, K, K)) * np.sqrt(2.0 / (K * K * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=pad)[0]
    OH, OW = y.shape[1:]
    res = rng.normal(size=(Cout, OH, OW)).astype(np.float32) if residual else None
    if residual:
        y = y + torch.from_numpy(res)
    want = (F.relu(y) if relu else y).numpy()
    d_w = dev.empty((w.size,), fill=np.nan)
    dev.call("mnc_pack_conv_weights", dev.put(w), d_w, Cout, Cin, K, K)
    d_y = dev.empty((Cout * OH * OW,), fill=np.nan)
    dev.call("mnc_conv2d", dev.put(to_c8(x)), d_w, dev.put(b), dev.put(to_c8(res)) if residual else None, d_y, H, W, Cin, Cout,
             K, K, stride, pad, relu)
    got = from_c8(dev.get(d_y, (Cout * OH * OW,)), Cout, OH, OW)
    assert not np.isnan(got).any()
    d, rel = err(got, want)
    assert rel < 1e-4, (d, rel)


@pytest.mark.parametrize("H,W,Cin,Cout,K,stride,pad,residual", GEN_CONV)
def test_conv2d_general_f16(dev, H, W, Cin, Cout, K, stride, pad, residual):
    """mnc_conv2d_f16: exact against torch on fp16-rounded operands (fp32 accumulation), ~3e-4 of range against fp32."""
    rng = np.random.default_rng(H * 100 + W + K + 1)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, K, K)) * np.sqrt(2.0 / (K * K * Cin))).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)

    def ref(xx, ww):
        y = F.conv2d(torch.from_numpy(xx)[None], torch.from_numpy(ww), torch.from_numpy(b), stride=stride, padding=pad)[0]
        return y
    OH, OW = ref(x, w).shape[1:]
    res = rng.normal(size=(Cout, OH, OW)).astype(np.float32) if residual else None
    fin = lambda y: F.relu(y + torch.from_numpy(res) if residual else y).numpy()
    d_w = dev.empty((K * K * ((Cin + 31) // 32) * 32 * Cout // 2,), fill=np.nan)
    dev.call("mnc_pack_conv_weights_f16", dev.put(w), d_w, Cout, Cin, K, K)
    d_y = dev.empty((Cout * OH * OW,), fill=np.nan)
    dev.call("mnc_conv2d_f16", dev.put(to_c8(x)), d_w, dev.put(b), dev.put(to_c8(res)) if residual else None, d_y, H, W, Cin,
             Cout, K, K, stride, pad, 1)
    got = from_c8(dev.get(d_y, (Cout * OH * OW,)), Cout, OH, OW)
    assert not np.isnan(got).any()
    rel = err(got, fin(ref(r16(x), r16(w))))[1]
    rel32 = err(got, fin(ref(x, w)))[1]
    print("conv2d f16 %dx%d %d->%d k%d s%d: vs fp16-rounded rel=%.3e, vs fp32 rel=%.3e" % (H, W, Cin, Cout, K, stride, rel, rel32))
    assert rel < 1e-5 and rel32 < 2e-3


# 1x1 GEMM convolution (csrc/conv1x1.hip): H, W, Cin, Cout, stride, residual.  Ragged pixel counts, channel counts that leave the
# last 32-channel tile / channel group partly empty, K-step counts 1, 2, 3, 5 (the 4-deep prefetch's phantom steps), the ResNet-50
# C4 shapes at 800x1333, and both strides.
C11 = [(13, 17, 16, 8, 1, False), (13, 17, 16, 24, 1, True), (20, 33, 32, 72, 2, True), (31, 45, 48, 64, 1, False),
       (31, 45, 80, 160, 2, True), (50, 84, 256, 1024, 1, True), (100, 167, 512, 256, 2, False), (200, 334, 64, 256, 1, True),
       (200, 334, 256, 64, 1, False), (100, 167, 128, 512, 1, True)]


@pytest.mark.parametrize("H,W,Cin,Cout,stride,residual", C11)
@pytest.mark.parametrize("relu", [1, 0])
def test_conv1x1_fp32(dev, monkeypatch, H, W, Cin, Cout, stride, residual, relu, tune):
    """mnc_conv1x1 (fp32 matrix pipe, operands straight from the c8 tensors) against torch fp32, default tile and two forced ones."""
    rng = np.random.default_rng(H * 100 + W + Cin)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 1, 1)) * np.sqrt(2.0 / Cin)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), stride=stride)[0]
    OH, OW = y.shape[1:]
    res = rng.normal(size=(Cout, OH, OW)).astype(np.float32) if residual else None
    if residual:
        y = y + torch.from_numpy(res)
    want = (F.relu(y) if relu else y).numpy()
    d_w = dev.empty((Cin * ((Cout + 31) // 32) * 32,), fill=np.nan)
    dev.call("mnc_pack_conv1x1", dev.put(w), d_w, Cout, Cin, 0)
    d_x, d_b, d_r = dev.put(to_c8(x)), dev.put(b), dev.put(to_c8(res)) if residual else None
    for tile in (None, "4,2", "1,1", "2,1"):
        if tile:
            tune("CONV1X1_TILE", tile)
        d_y = dev.empty((Cout * OH * OW,), fill=np.nan)
        dev.call("mnc_conv1x1", d_x, d_w, d_b, d_r, d_y, H, W, Cin, Cout, stride, relu)
        got = from_c8(dev.get(d_y, (Cout * OH * OW,)), Cout, OH, OW)
        assert not np.isnan(got).any(), tile
        d, rel = err(got, want)
        assert rel < 1e-4, (tile, d, rel)


@pytest.mark.parametrize("H,W,Cin,Cout,stride,residual", C11)
@pytest.mark.parametrize("res_packed,out_packed", [(1, 1), (0, 0), (1, 0), (0, 1)])
def test_conv1x1_f16_packed(dev, monkeypatch, H, W, Cin, Cout, stride, residual, res_packed, out_packed, tune):
    """mnc_conv1x1_f16_pk: packed fp16 c8 activations in, packed fp16 or fp32 out, residual in either form.  Against torch on the
    same fp16-rounded operands the fp32 result agrees to accumulation order (1e-5 of range); a packed output is that result
    rounded to fp16 once (half an fp16 ulp of the value on top)."""
    if not residual and res_packed:
        pytest.skip("no residual")
    rng = np.random.default_rng(H * 100 + W + Cin + 7)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    x = r16(rng.normal(size=(Cin, H, W)).astype(np.float32))
    w = (rng.normal(size=(Cout, Cin, 1, 1)) * np.sqrt(2.0 / Cin)).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    y = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(r16(w)), torch.from_numpy(b), stride=stride)[0]
    OH, OW = y.shape[1:]
    res = rng.normal(size=(Cout, OH, OW)).astype(np.float32) if residual else None
    if residual and res_packed:
        res = r16(res)
    if residual:
        y = y + torch.from_numpy(res)
    want = F.relu(y).numpy()
    d_w = dev.empty((Cin * ((Cout + 31) // 32) * 32 // 2,), fill=np.nan)
    dev.call("mnc_pack_conv1x1", dev.put(w), d_w, Cout, Cin, 1)
    d_x = dev.put(to_c8(x).astype(np.float16), dtype=np.float16)
    d_r = None
    if residual:
        d_r = dev.put(to_c8(res).astype(np.float16), dtype=np.float16) if res_packed else dev.put(to_c8(res))
    for tile in (None, "4,2", "1,2"):
        if tile:
            tune("CONV1X1_TILE", tile)
        n = Cout * OH * OW
        d_y = dev.empty((n,), fill=np.nan)
        dev.call("mnc_conv1x1_f16_pk", d_x, d_w, dev.put(b), d_r, d_y, H, W, Cin, Cout, stride, 1, res_packed, out_packed)
        if out_packed:
            got = from_c8(dev.get(d_y, (n,), dtype=np.float16).astype(np.float32), Cout, OH, OW)
            assert not np.isnan(got).any(), tile
            # the fp32 result rounded once: within half an fp16 ulp (2^-11 relative, 2^-25 absolute floor) + the fp32 slack
            tol = np.abs(want) * 2.0 ** -11 + 2.0 ** -24 + 2e-5 * np.abs(want).max()
            assert (np.abs(got - want) <= tol).all(), (tile, float(np.abs(got - want).max()))
        else:
            got = from_c8(dev.get(d_y, (n,)), Cout, OH, OW)
            assert not np.isnan(got).any(), tile
            d, rel = err(got, want)
            assert rel < 1e-5, (tile, d, rel)


@pytest.mark.parametrize("H,W,K,stride,pad", [(112, 112, 3, 2, 0), (101, 166, 3, 2, 1), (37, 41, 2, 2, 0)])
def test_packed_fp16_stem_and_general_maxpool(dev, H, W, K, stride, pad):
    """The f16 mode's 2-byte tensors outside the MFMA kernels: the stem convolution's packed output is its fp32 output rounded to
    fp16 (nearest even), MAX pooling on the packed tensor is bit for bit the fp16 form of pooling the fp32 tensor."""
    rng = np.random.default_rng(H + K)
    x = rng.normal(size=(16, H, W)).astype(np.float32).astype(np.float16)
    want = F.max_pool2d(torch.from_numpy(x.astype(np.float32))[None], K, stride, pad, ceil_mode=True)[0].numpy()
    OH, OW = want.shape[1:]
    d_y = dev.empty((16 * OH * OW,), dtype=np.float16, fill=np.nan)
    dev.call("mnc_maxpool_c8_f16", dev.put(to_c8(x), dtype=np.float16), d_y, 16, H, W, K, stride, pad)
    got = from_c8(dev.get(d_y, (16 * OH * OW,), dtype=np.float16), 16, OH, OW)
    assert np.array_equal(got.astype(np.float32), want)
    im = rng.uniform(-120, 130, (3, H, W)).astype(np.float32)
    w = (rng.normal(size=(64, 3, 7, 7)) * 0.01).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    SH, SW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    d_a, d_h = dev.empty((64 * SH * SW,), fill=np.nan), dev.empty((64 * SH * SW,), dtype=np.float16, fill=np.nan)
    dev.call("mnc_conv_stem_c3", dev.put(im), dev.put(w), dev.put(b), d_a, H, W, 64, 7, 2, 3, 1)
    dev.call("mnc_conv_stem_c3_fmt", dev.put(im), dev.put(w), dev.put(b), d_h, H, W, 64, 7, 2, 3, 1, 1)
    a = dev.get(d_a, (64 * SH * SW,))
    assert np.array_equal(dev.get(d_h, (64 * SH * SW,), dtype=np.float16), a.astype(np.float16))


@pytest.mark.parametrize("H,W,K,stride,pad", [(75, 101, 7, 2, 3), (64, 64, 7, 2, 3), (31, 45, 3, 1, 1)])
def test_conv_stem_c3(dev, H, W, K, stride, pad):
    rng = np.random.default_rng(H + W)
    x = rng.uniform(-120, 130, (3, H, W)).astype(np.float32)
    w = (rng.normal(size=(64, 3, K, K)) * 0.01).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    want = F.relu(F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=pad))[0].numpy()
    OH, OW = want.shape[1:]
    d_y = dev.empty((64 * OH * OW,), fill=np.nan)
    dev.call("mnc_conv_stem_c3", dev.put(x), dev.put(w), dev.put(b), d_y, H, W, 64, K, stride, pad, 1)
    got = from_c8(dev.get(d_y, (64 * OH * OW,)), 64, OH, OW)
    assert err(got, want)[1] < 1e-5


@pytest.mark.parametrize("H,W,K,stride,pad,Cout", [(75, 101, 7, 2, 3, 64), (64, 64, 7, 2, 3, 64), (31, 45, 3, 1, 1, 32),
                                                    (40, 150, 5, 2, 2, 96), (9, 70, 7, 2, 3, 64), (200, 333, 7, 2, 3, 64)])
@pytest.mark.parametrize("out_packed", [0, 1])
def test_conv_stem_f16_mfma(dev, H, W, K, stride, pad, Cout, out_packed):
    """mnc_conv_stem_f16 (the stem as a GEMM on the fp16 matrix pipe, B fragments straight from the NCHW blob): exact against torch
    on fp16-rounded operands up to accumulation order, ~3e-4 of range against fp32; image borders, row segments that end inside
    a 32-pixel tile, maps narrower than a tile's windows (every segment takes the border path), odd sizes."""
    rng = np.random.default_rng(H + W + K)
    x = rng.uniform(-120, 130, (3, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, 3, K, K)) * 0.02).astype(np.float32)
    b = rng.normal(size=Cout).astype(np.float32)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    ref = lambda xx, ww: F.relu(F.conv2d(torch.from_numpy(xx)[None], torch.from_numpy(ww), torch.from_numpy(b), stride=stride,
                                         padding=pad))[0].numpy()
    want16, want32 = ref(r16(x), r16(w)), ref(x, w)
    OH, OW = want32.shape[1:]
    KS = (3 * K + 1) // 2
    d_w = dev.empty((KS * (Cout // 32) * 1024,), dtype=np.uint8, fill=0xFF)
    dev.call("mnc_pack_conv_stem_f16", dev.put(w), d_w, Cout, K)
    n = Cout * OH * OW
    d_y = dev.empty((n,), fill=np.nan)
    dev.call("mnc_conv_stem_f16", dev.put(x), d_w, dev.put(b), d_y, H, W, Cout, K, stride, pad, 1, out_packed)
    if out_packed:
        got = from_c8(dev.get(d_y, (n,), dtype=np.float16).astype(np.float32), Cout, OH, OW)
        assert not np.isnan(got).any()
        tol = np.abs(want16) * 2.0 ** -11 + 2.0 ** -24 + 2e-5 * np.abs(want16).max()
        assert (np.abs(got - want16) <= tol).all(), float(np.abs(got - want16).max())
    else:
        got = from_c8(dev.get(d_y, (n,)), Cout, OH, OW)
        assert not np.isnan(got).any()
        rel, rel32 = err(got, want16)[1], err(got, want32)[1]
        print("stem f16 %dx%d k%d s%d -> %d: vs fp16-rounded rel=%.3e, vs fp32 rel=%.3e" % (H, W, K, stride, Cout, rel, rel32))
        assert rel < 1e-5 and rel32 < 2e-3


@pytest.mark.parametrize("H,W,K,stride,pad", [(112, 112, 3, 2, 0), (100, 167, 3, 2, 0), (101, 166, 3, 2, 1), (37, 41, 2, 2, 0)])
def test_maxpool_general_and_add(dev, H, W, K, stride, pad):
    rng = np.random.default_rng(H)
    x = rng.normal(size=(16, H, W)).astype(np.float32)
    want = F.max_pool2d(torch.from_numpy(x)[None], K, stride, pad, ceil_mode=True)[0].numpy()      # Caffe's pooling size rule
    OH, OW = want.shape[1:]
    d_y = dev.empty((16 * OH * OW,), fill=np.nan)
    dev.call("mnc_maxpool_c8", dev.put(to_c8(x)), d_y, 16, H, W, K, stride, pad)
    assert np.array_equal(from_c8(dev.get(d_y, (16 * OH * OW,)), 16, OH, OW), want)
    a, b = rng.normal(size=1003).astype(np.float32), rng.normal(size=1003).astype(np.float32)
    for relu in (0, 1):
        d_o = dev.empty((1003,), fill=np.nan)
        dev.call("mnc_add", dev.put(a), dev.put(b), d_o, 1003, relu)
        assert np.array_equal(dev.get(d_o, (1003,)), np.maximum(a + b, 0) if relu else a + b)


def _rois(rng, R, W, H):
    b = GI._boxes(rng, R, W, H, 8, 500)
    b[0] = [0, 0, W - 1, H - 1]                  # whole image
    b[1] = [10.3, 20.7, 10.9, 21.2]              # sub-pixel
    b[2] = [W - 1, H - 1, W - 1, H - 1]          # bottom-right corner
    b[3] = [200, 100, 150, 60]                   # malformed x2 < x1
    b[4] = [0, 0, 15, 15]
    return np.hstack([np.zeros((R, 1), np.float32), b]).astype(np.float32)


@pytest.mark.parametrize("pool2,P", [(0, 14), (1, 14), (0, 7)])
@pytest.mark.parametrize("variant", [None, "1", "4", "3"])
def test_roi_warp(dev, monkeypatch, pool2, P, variant, tune):
    """(variant: the library's choice by channel count, or MNC_ROI_WARP_VARIANT forcing the one-wave-per-position / the
    4-channels-per-thread / the one-wave-per-output-row kernel -- all bit-exact with the oracle, rois partly and wholly outside the
    map included.)"""
    if variant:
        tune("ROI_WARP_VARIANT", variant)
    rng = np.random.default_rng(4)
    C, H, W, R = 64, 38, 63, 40
    feat = rng.normal(size=(C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 1000, 600)
    d_out = dev.empty((R * P * P * C,), fill=np.nan)
    dev.call("mnc_roi_warp", dev.put(to_c8(feat)), C, H, W, dev.put(rois), R, P, P, 0.0625, pool2, d_out)
    got = dev.get(d_out, (R, P, P, C)).transpose(0, 3, 1, 2)
    if pool2:
        want = native.maxpool2(native.roi_warp(feat, rois, 2 * P, 2 * P, 0.0625))
    else:
        want = native.roi_warp(feat, rois, P, P, 0.0625)
    assert np.array_equal(got, want)      # roi.hip is built with -ffp-contract=off and mirrors the oracle op by op


@pytest.mark.parametrize("pool2", [0, 1])
@pytest.mark.parametrize("C", [512, 320])
def test_roi_warp_row_kernel_at_head_width(dev, pool2, C, tune):
    """ROI_WARP_VARIANT=3: one wave per output row, the feature-map columns of neighbouring positions kept in registers.  512
    channels (two 256-channel pieces per row) and 320 (a ragged second piece), 120 rois of every kind -- proposal-sized, thin, tiny
    (all samples in one cell), wider than 28 cells (samples more than a cell apart), touching every border, outside the map:
    bit-exact with the oracle and with the one-wave-per-position kernel."""
    rng = np.random.default_rng(31 + pool2 + C)
    H, W, R, P = 38, 63, 120, 14
    feat = rng.normal(size=(C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 1000, 600)
    rois[5] = [0, 0, 0, 999, 40]                 # full width, thin: 63 cells over 14 / 28 samples
    rois[6] = [0, 980, 560, 999, 599]            # bottom-right corner region
    rois[7] = [0, 0, 0, 5, 5]                    # inside one cell
    rois[8] = [0, -50, -30, 80, 70]              # partly left / above the map
    rois[9] = [0, 300.5, 200.25, 310.75, 590.0]  # thin and tall
    d_out = dev.empty((R * P * P * C,), fill=np.nan)
    d_feat, d_rois = dev.put(to_c8(feat)), dev.put(rois)
    res = {}
    for v in ("3", "1"):
        tune("ROI_WARP_VARIANT", v)
        dev.put_into(d_out, np.full((R * P * P * C,), np.nan, np.float32))
        dev.call("mnc_roi_warp", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, d_out)
        res[v] = dev.get(d_out, (R, P, P, C)).copy()
    assert np.array_equal(res["3"], res["1"])
    want = native.maxpool2(native.roi_warp(feat, rois, 2 * P, 2 * P, 0.0625)) if pool2 else native.roi_warp(feat, rois, P, P, 0.0625)
    assert np.array_equal(res["3"].transpose(0, 3, 1, 2), want)


@pytest.mark.parametrize("fmt", [1, 2, 3])
@pytest.mark.parametrize("variant", [None, "4", "8"])
def test_per_roi_producers_write_the_fc_activation_form(dev, monkeypatch, fmt, variant, tune):
    """(variant: the library's choice, or MNC_ROI_SM_VARIANT forcing the 4- / 8-channels-per-thread kernels.)
    mnc_roi_warp_sm / mnc_maxpool2_rhwc_sm / mnc_mask_pool_sm: the fp32 output is bit for bit that of the plain entry point, and
    the second output is bit for bit what mnc_fc_pack_act (the InnerProduct's own conversion pass) makes of it -- fp16 stage-major
    (fmt 1) and split bf16 stage-major (fmt 2)."""
    if variant:
        tune("ROI_SM_VARIANT", variant)
        tune("ROI_WARP_VARIANT", {"4": "4", "8": "8"}[variant])
    rng = np.random.default_rng(40 + fmt)
    C, H, W, R, P = 64, 38, 63, 37, 14
    feat = rng.normal(size=(C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 1000, 600)
    d_feat, d_rois = dev.put(to_c8(feat)), dev.put(rois)
    eb = 4 if fmt == 2 else 2
    pack_flag = {1: 1, 2: 0, 3: 2}                   # mnc_fc_pack_act's argument for stage-major format 1 / 2 / 3

    def shadow_of(d_rows, M, K):
        d = dev.empty((M * K * eb,), dtype=np.uint8, fill=0)
        dev.call("mnc_fc_pack_act", d_rows, d, M, K, pack_flag[fmt])
        return dev.get(d, (M * K * eb,), dtype=np.uint8)

    for pool2, wave in ((0, False), (1, False), (0, True), (1, True)):
        if wave:
            if variant:
                continue
            tune("ROI_WARP_VARIANT", "1")         # the one-wave-per-position kernel writes it too
        K = P * P * C
        d_a, d_b = dev.empty((R * K,), fill=np.nan), dev.empty((R * K,), fill=np.nan)
        d_sm = dev.empty((R * K * eb,), dtype=np.uint8, fill=0xAB)
        dev.call("mnc_roi_warp", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, d_a)
        dev.call("mnc_roi_warp_sm", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, d_b, d_sm, fmt)
        assert np.array_equal(dev.get(d_a, (R * K,)), dev.get(d_b, (R * K,)))
        assert np.array_equal(dev.get(d_sm, (R * K * eb,), dtype=np.uint8), shadow_of(d_b, R, K)), ("roi_warp", pool2, wave)
    dev.tune("ROI_WARP_VARIANT", None)
    # Pooling and MaskPooling on the 14x14 tensor
    x = rng.normal(size=(R, P, P, C)).astype(np.float32)
    m = rng.uniform(0, 1, (R, P, P)).astype(np.float32)
    d_x, d_m = dev.put(x), dev.put(m)
    K7 = 49 * C
    d_a, d_b = dev.empty((R * K7,), fill=np.nan), dev.empty((R * K7,), fill=np.nan)
    d_sm = dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0xAB)
    dev.call("mnc_maxpool2_rhwc", d_x, d_a, R, P, P, C)
    dev.call("mnc_maxpool2_rhwc_sm", d_x, d_b, R, P, P, C, d_sm, fmt)
    assert np.array_equal(dev.get(d_a, (R * K7,)), dev.get(d_b, (R * K7,)))
    assert np.array_equal(dev.get(d_sm, (R * K7 * eb,), dtype=np.uint8), shadow_of(d_b, R, K7))
    for pool2, K in ((1, K7), (0, P * P * C)):
        d_a, d_b = dev.empty((R * K,), fill=np.nan), dev.empty((R * K,), fill=np.nan)
        d_sm = dev.empty((R * K * eb,), dtype=np.uint8, fill=0xAB)
        dev.call("mnc_mask_pool", d_x, d_m, d_a, R, P, P, C, pool2)
        dev.call("mnc_mask_pool_sm", d_x, d_m, d_b, R, P, P, C, pool2, d_sm, fmt)
        assert np.array_equal(dev.get(d_a, (R * K,)), dev.get(d_b, (R * K,)))
        assert np.array_equal(dev.get(d_sm, (R * K * eb,), dtype=np.uint8), shadow_of(d_b, R, K)), ("mask_pool", pool2)
    # both poolings of the tensor in one pass (mnc_box_mask_pool): the two outputs and their second outputs, bit for bit
    d_box, d_mk = dev.empty((R * K7,), fill=np.nan), dev.empty((R * K7,), fill=np.nan)
    dev.call("mnc_maxpool2_rhwc", d_x, d_box, R, P, P, C)
    dev.call("mnc_mask_pool", d_x, d_m, d_mk, R, P, P, C, 1)
    for f in (0, fmt):
        d_b2, d_m2 = dev.empty((R * K7,), fill=np.nan), dev.empty((R * K7,), fill=np.nan)
        d_bsm = dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0xAB)
        d_msm = dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0xAB)
        dev.call("mnc_box_mask_pool", d_x, d_m, d_b2, d_m2, R, P, P, C, d_bsm if f else None, d_msm if f else None, f)
        assert np.array_equal(dev.get(d_box, (R * K7,)), dev.get(d_b2, (R * K7,))), f
        assert np.array_equal(dev.get(d_mk, (R * K7,)), dev.get(d_m2, (R * K7,))), f
        if f:
            assert np.array_equal(dev.get(d_bsm, (R * K7 * eb,), dtype=np.uint8), shadow_of(d_b2, R, K7))
            assert np.array_equal(dev.get(d_msm, (R * K7 * eb,), dtype=np.uint8), shadow_of(d_m2, R, K7))
    # round 6 (mnc_box_mask_pool_ex): the fp32 outputs omitted -> the same stage-major bits; the tensor read from its stage-major fp16
    # copy (format 1) -> exactly the fp32 pass on the fp16-ROUNDED tensor (box pool: the bits of the pass on the unrounded one)
    d_bs, d_ms = (dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0xAB) for _ in range(2))
    dev.call("mnc_box_mask_pool_ex", d_x, None, 0, d_m, None, None, R, P, P, C, d_bs, d_ms, fmt)
    assert np.array_equal(dev.get(d_bs, (R * K7 * eb,), dtype=np.uint8), dev.get(d_bsm, (R * K7 * eb,), dtype=np.uint8))
    assert np.array_equal(dev.get(d_ms, (R * K7 * eb,), dtype=np.uint8), dev.get(d_msm, (R * K7 * eb,), dtype=np.uint8))
    K14 = P * P * C
    for in_fmt in ((3,) if fmt == 3 else (1, 2)):    # (the bf16 form pairs with itself only)
        ib = 4 if in_fmt == 2 else 2
        d_xsm = dev.empty((R * K14 * ib,), dtype=np.uint8, fill=0xAB)
        dev.call("mnc_fc_pack_act", d_x, d_xsm, R, K14, pack_flag[in_fmt])
        if in_fmt == 1:
            xr = x.astype(np.float16).astype(np.float32)
        elif in_fmt == 3:
            xr = _rbf16(x)
        else:                                          # split bf16: hi = rne(x), lo = rne(x - hi); the form holds hi + lo
            hi = _rbf16(x)
            xr = hi + _rbf16(x - hi)
        d_xr = dev.put(xr)
        want = [dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0) for _ in range(2)]
        d_wb, d_wm = dev.empty((R * K7,), fill=np.nan), dev.empty((R * K7,), fill=np.nan)
        dev.call("mnc_box_mask_pool", d_xr, d_m, d_wb, d_wm, R, P, P, C, want[0], want[1], fmt)
        for with_f32 in (True, False):
            d_b3, d_m3 = dev.empty((R * K7,), fill=np.nan), dev.empty((R * K7,), fill=np.nan)
            d_bs, d_ms = (dev.empty((R * K7 * eb,), dtype=np.uint8, fill=0xAB) for _ in range(2))
            dev.call("mnc_box_mask_pool_ex", None, d_xsm, in_fmt, d_m, d_b3 if with_f32 else None, d_m3 if with_f32 else None, R, P, P, C,
                     d_bs, d_ms, fmt)
            assert np.array_equal(dev.get(d_bs, (R * K7 * eb,), dtype=np.uint8), dev.get(want[0], (R * K7 * eb,), dtype=np.uint8)), in_fmt
            assert np.array_equal(dev.get(d_ms, (R * K7 * eb,), dtype=np.uint8), dev.get(want[1], (R * K7 * eb,), dtype=np.uint8)), in_fmt
            if with_f32:
                assert np.array_equal(dev.get(d_b3, (R * K7,)), dev.get(d_wb, (R * K7,)))
                assert np.array_equal(dev.get(d_m3, (R * K7,)), dev.get(d_wm, (R * K7,)))
        if fmt == 1 and in_fmt == 1:
            assert np.array_equal(dev.get(want[0], (R * K7 * eb,), dtype=np.uint8), dev.get(d_bsm, (R * K7 * eb,), dtype=np.uint8))
    # the warp with its fp32 output omitted: the same stage-major bits (both kernels the SPEC's convention runs below 1024 channels)
    for pool2 in ((0, 1) if variant != "8" else ()):
        K = P * P * C
        d_f, d_s0, d_s1 = dev.empty((R * K,), fill=np.nan), dev.empty((R * K * eb,), dtype=np.uint8, fill=0xAB), dev.empty((R * K * eb,), dtype=np.uint8, fill=0xCD)
        dev.call("mnc_roi_warp_sm", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, d_f, d_s0, fmt)
        dev.call("mnc_roi_warp_sm", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, None, d_s1, fmt)
        assert np.array_equal(dev.get(d_s0, (R * K * eb,), dtype=np.uint8), dev.get(d_s1, (R * K * eb,), dtype=np.uint8)), pool2
    with pytest.raises(Exception):
        dev.call("mnc_roi_warp_sm", d_feat, C, H, W, d_rois, R, P, P, 0.0625, 0, None, None, 0)
    # a channel count whose 8-channel groups would straddle a stage is refused, not mis-packed
    with pytest.raises(Exception):
        dev.call("mnc_maxpool2_rhwc_sm", d_x, d_b, R, P, P, 24, d_sm, fmt)


@pytest.mark.parametrize("fmt", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(300, 512, 12544), (1000, 512, 12544), (700, 256, 6272), (37, 256, 12544), (160, 4096, 1024)])
def test_fc_on_prepacked_activations(dev, fmt, M, N, K):
    """mnc_fc_{f16,bf16x3}_pre on the stage-major activation tensor == mnc_fc_{f16,bf16x3} on the fp32 rows, bit for bit, for
    one row block, several row blocks with a ragged tail (two launches sharing one tensor: m_stride > M), the small-GEMM tile
    and the 160-row tile; and on a tensor that holds more rows than the call multiplies."""
    rng = np.random.default_rng(M + N + fmt)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) * 0.02).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    eb = 4 if fmt == 2 else 2
    name = {1: "f16", 2: "bf16x3", 3: "bf16"}[fmt]           # (3, round 6: the plain bf16 mode's form -- fp16's layout, bf16 values)
    pack_flag = {1: 1, 2: 0, 3: 2}
    d_w = dev.empty(((N + 127) // 128 * 128 * K * eb,), dtype=np.uint8)
    dev.call("mnc_pack_fc_" + name, dev.put(w), d_w, N, K)
    d_a, d_b = dev.put(a), dev.put(b)
    d_sm = dev.empty((M * K * eb,), dtype=np.uint8)
    dev.call("mnc_fc_pack_act", d_a, d_sm, M, K, pack_flag[fmt])
    d_y0, d_y1 = dev.empty((M * N,), fill=np.nan), dev.empty((M * N,), fill=np.nan)
    dev.call("mnc_fc_" + name, d_a, d_w, d_b, d_y0, M, N, K, N, 1)

    def pre(d_out, rows):
        if fmt == 3:
            dev.call("mnc_fc_bf16_ex", None, d_sm, M, d_w, d_b, d_out, rows, N, K, N, 1, None, 0)
        else:
            dev.call("mnc_fc_%s_pre" % name, d_sm, M, d_w, d_b, d_out, rows, N, K, N, 1)
    pre(d_y1, M)
    y0 = dev.get(d_y0, (M, N))
    assert not np.isnan(y0).any() and np.array_equal(y0, dev.get(d_y1, (M, N)))
    # the general entry point: either input form, and the result rows a second time in the NEXT InnerProduct's form (written by
    # the K-split reduction, or converted from the stored rows when there is a single split) == mnc_fc_pack_act of the output
    if N % 64 == 0:
        for ofmt in ((3,) if fmt == 3 else (1, 2)):
            oeb = 4 if ofmt == 2 else 2
            d_y4, d_osm = dev.empty((M * N,), fill=np.nan), dev.empty((M * N * oeb,), dtype=np.uint8, fill=0xCD)
            dev.call("mnc_fc_%s_ex" % name, None, d_sm, M, d_w, d_b, d_y4, M, N, K, N, 1, d_osm, ofmt)
            assert np.array_equal(y0, dev.get(d_y4, (M, N)))
            d_ref = dev.empty((M * N * oeb,), dtype=np.uint8, fill=0)
            dev.call("mnc_fc_pack_act", d_y4, d_ref, M, N, pack_flag[ofmt])
            assert np.array_equal(dev.get(d_osm, (M * N * oeb,), dtype=np.uint8), dev.get(d_ref, (M * N * oeb,), dtype=np.uint8)), ofmt
        d_y5 = dev.empty((M * N,), fill=np.nan)
        dev.call("mnc_fc_%s_ex" % name, d_a, None, 0, d_w, d_b, d_y5, M, N, K, N, 1, None, 0)
        assert np.array_equal(y0, dev.get(d_y5, (M, N)))
    # mnc_fc_unpack_act (round 6): the rows back out of the stage-major tensor = the fp32 rows rounded to the form
    d_back = dev.empty((M * K,), fill=np.nan)
    dev.call("mnc_fc_unpack_act", d_sm, d_back, M, K, fmt)
    back = dev.get(d_back, (M, K))
    if fmt == 1:
        assert np.array_equal(back, a.astype(np.float16).astype(np.float32))
    elif fmt == 3:
        assert np.array_equal(back, _rbf16(a))
    else:
        hi = _rbf16(a)
        assert np.array_equal(back, hi + _rbf16(a - hi))
    if M >= 300:        # the first 2/3 of the rows of the same tensor (m_stride = M > rows multiplied)
        Mh = M * 2 // 3
        d_y2, d_y3 = dev.empty((Mh * N,), fill=np.nan), dev.empty((Mh * N,), fill=np.nan)
        dev.call("mnc_fc_" + name, d_a, d_w, d_b, d_y2, Mh, N, K, N, 1)
        pre(d_y3, Mh)
        assert np.array_equal(dev.get(d_y2, (Mh, N)), dev.get(d_y3, (Mh, N)))


@pytest.mark.parametrize("P,N", [(7, 1), (14, 3)])
def test_roi_pool(dev, P, N):
    """ROIPooling (CFM graph) over a batch of c8 images: bit-exact with the oracle (pure max / integer bin arithmetic),
    including one-pixel, partly-outside, fully-outside (empty bins -> 0) rois and a NaN feature (skipped by `v > max`)."""
    rng = np.random.default_rng(14)
    C, H, W, R = 64, 38, 63, 60
    feat = rng.normal(size=(N, C, H, W)).astype(np.float32)
    feat[0, 3, 5, 7] = np.nan
    rois = np.hstack((rng.integers(0, N, (R, 1)).astype(np.float32), _rois(rng, R, 1000, 600)[:, 1:]))
    rois[0, 1:] = [80, 80, 80, 80]
    rois[1, 1:] = [1100, 700, 1300, 900]
    rois[2, 1:] = [900, 500, 1200, 800]
    rois[3] = [0, 100, 70, 130, 100]
    d_out = dev.empty((R * P * P * C,), fill=np.nan)
    c8 = np.stack([to_c8(f) for f in feat])
    dev.call("mnc_roi_pool", dev.put(c8), N, C, H, W, dev.put(rois), R, P, P, 0.0625, d_out)
    got = dev.get(d_out, (R, P, P, C)).transpose(0, 3, 1, 2)
    want = native.roi_pool(feat, rois, P, P, 0.0625)
    assert np.array_equal(got, want) and not np.isnan(got).any()
    assert not got[1].any() and got[2].any()
    # full-map roi at 1x1: the global max of each channel (property at the real conv5_3 size)
    whole = np.array([[N - 1, 0, 0, (W - 1) * 16, (H - 1) * 16]], np.float32)
    d_one = dev.empty((C,), fill=np.nan)
    dev.call("mnc_roi_pool", dev.put(c8), N, C, H, W, dev.put(whole), 1, 1, 1, 0.0625, d_one)
    assert np.array_equal(dev.get(d_one, (C,)), np.nanmax(feat[N - 1].reshape(C, -1), axis=1))


def test_roi_warp_interior_is_plain_bilinear(dev):
    """SPEC.md 1 property: a RoI whose sample grid lands exactly on feature-map pixels reproduces them."""
    C, H, W = 8, 20, 20
    feat = np.random.default_rng(5).normal(size=(C, H, W)).astype(np.float32)
    # x1s = 2, x2s = 2 + 7 - 1 -> roi_w = 7 = bin count -> samples at 2, 3, ..., 8
    rois = np.array([[0, 32, 48, 32 + 6 * 16, 48 + 6 * 16]], np.float32)
    d_out = dev.empty((7 * 7 * C,), fill=np.nan)
    dev.call("mnc_roi_warp", dev.put(to_c8(feat)), C, H, W, dev.put(rois), 1, 7, 7, 0.0625, 0, d_out)
    got = dev.get(d_out, (7, 7, C)).transpose(2, 0, 1)
    assert np.array_equal(got, feat[:, 3:10, 2:9])


def test_roi_elementwise_ops(dev):
    rng = np.random.default_rng(6)
    R, C = 37, 64
    feat = rng.normal(size=(R, C, 14, 14)).astype(np.float32)
    mask = rng.uniform(0, 1, (R, 1, 21, 21)).astype(np.float32)
    hwc = np.ascontiguousarray(feat.transpose(0, 2, 3, 1))
    # MaskResize 21 -> 14
    d_m14 = dev.empty((R * 196,), fill=np.nan)
    dev.call("mnc_mask_resize", dev.put(mask), d_m14, R, 21, 21, 14, 14)
    m14 = dev.get(d_m14, (R, 1, 14, 14))
    want14 = native.mask_resize(mask, 14, 14)
    assert err(m14, want14)[0] < 1e-6
    # MaskPooling (+ fused pool)
    d_feat = dev.put(hwc)
    d_o = dev.empty((R * 196 * C,), fill=np.nan)
    dev.call("mnc_mask_pool", d_feat, d_m14, d_o, R, 14, 14, C, 0)
    got = dev.get(d_o, (R, 14, 14, C)).transpose(0, 3, 1, 2)
    want = native.mask_pool(feat, m14)
    assert np.array_equal(got, want)
    d_o2 = dev.empty((R * 49 * C,), fill=np.nan)
    dev.call("mnc_mask_pool", d_feat, d_m14, d_o2, R, 14, 14, C, 1)
    got2 = dev.get(d_o2, (R, 7, 7, C)).transpose(0, 3, 1, 2)
    assert np.array_equal(got2, native.maxpool2(want))
    # Pooling on per-RoI features
    d_o3 = dev.empty((R * 49 * C,), fill=np.nan)
    dev.call("mnc_maxpool2_rhwc", d_feat, d_o3, R, 14, 14, C)
    assert np.array_equal(dev.get(d_o3, (R, 7, 7, C)).transpose(0, 3, 1, 2), native.maxpool2(feat))
    # layout round trips
    d_a, d_b = dev.empty((R * C * 196,)), dev.empty((R * C * 196,))
    dev.call("mnc_rchw_to_rhwc", dev.put(feat), d_a, R, C, 14, 14)
    assert np.array_equal(dev.get(d_a, hwc.shape), hwc)
    dev.call("mnc_rhwc_to_rchw", d_a, d_b, R, C, 14, 14)
    assert np.array_equal(dev.get(d_b, feat.shape), feat)
    x = rng.normal(size=(16, 9, 11)).astype(np.float32)
    d_c, d_d = dev.empty((x.size,)), dev.empty((x.size,))
    dev.call("mnc_nchw_to_c8", dev.put(x), d_c, 16, 9, 11)
    assert np.array_equal(dev.get(d_c, to_c8(x).shape), to_c8(x))
    dev.call("mnc_c8_to_nchw", d_c, d_d, 16, 9, 11)
    assert np.array_equal(dev.get(d_d, x.shape), x)


# (M, N, K, act): split-K and fused paths, partial row/column tiles, the real head shapes
FC_SHAPES = [(300, 4096, 4096, 1), (45, 150, 64, 0), (1, 441, 256, 2), (300, 441, 256, 2), (300, 126, 8192, 0),
             (300, 256, 14 * 14 * 512, 1), (7, 4096, 25088, 1), (321, 128, 512, 0),
             (1000, 512, 2048, 1), (760, 1024, 1536, 0), (640, 512, 3200, 2)]     # several row blocks: full blocks + ragged tail


@pytest.mark.parametrize("M,N,K,act", FC_SHAPES)
def test_fc_mfma(dev, M, N, K, act):
    rng = np.random.default_rng(M + N + K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_o = dev.empty((M * N,), fill=np.nan)
    dev.call("mnc_fc", dev.put(a), dev.put(w), dev.put(b), d_o, M, N, K, N, act)
    got = dev.get(d_o, (M, N))
    y = F.linear(torch.from_numpy(a), torch.from_numpy(w), torch.from_numpy(b))
    want = (F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y).numpy()
    assert not np.isnan(got).any()
    d, rel = err(got, want)
    assert rel < 1e-4, (d, rel)


@pytest.mark.parametrize("M,N,K,act,ldc_pad", [(300, 4096, 25088, 1, 0), (300, 520, 4096, 1, 8), (290, 1024, 2112, 0, 0),
                                               (640, 256, 6400, 1, 0), (161, 128, 8192, 2, 64), (304, 384, 8192, 0, 0),
                                               (289, 1000, 4096, 2, 24)])
def test_fc_mfma_lds_dma(dev, monkeypatch, M, N, K, act, ldc_pad, tune):
    """fc_mfma_dma_kernel (320-row blocks, operand panels copied global -> LDS by DMA into XOR-swizzled rows; fc6's kernel):
    against torch, and against the register-staged kernel (MNC_FC_DMA=0) on the same call -- the products and the order inside a
    K split are the same, the split boundaries differ (even stage counts).  MNC_FC_TILE=10 puts shapes on it that the tile
    heuristic would give to the 160-row kernel: N that does not fill the last column tile, rows that do not fill the block,
    two row blocks, a K with an odd number of 64-deep steps, a column slice (ldc > N), sigmoid."""
    rng = np.random.default_rng(M + N + K + 5)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_a, d_w, d_b = dev.put(a), dev.put(w), dev.put(b)
    ld = N + ldc_pad
    y = F.linear(torch.from_numpy(a), torch.from_numpy(w), torch.from_numpy(b))
    want = (F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y).numpy()
    tune("FC_TILE", "10")

    def run():
        d_o = dev.empty((M * ld,), fill=np.nan)
        dev.call("mnc_fc", d_a, d_w, d_b, d_o, M, N, K, ld, act)
        got = dev.get(d_o, (M, ld))
        assert not np.isnan(got[:, :N]).any() and np.isnan(got[:, N:]).all()
        d, rel = err(got[:, :N], want)
        assert rel < 1e-4, (d, rel)
        return got[:, :N]

    # the product build (eight waves, 16 x 16 x 4 fragments) and the register-staged kernel: same products, other order
    tune("FC_DMA", "1")
    prod = run()
    tune("FC_DMA", "0")
    staged = run()
    assert err(prod, staged)[1] < 1e-5
    if not TUNING_BUILD:
        return
    # tuning builds: the 32 x 32 x 2 DMA kernel the product build replaced -- eight waves, four waves, with and without its half tile
    tune("FC_DMA", "1")
    tune("FC_MFMA16", "0")
    w8 = run()
    tune("FC_DMA_WAVES", "4")
    w4 = run()
    dev.tune("FC_DMA_WAVES", None)
    assert np.array_equal(w8, w4) and err(w8, staged)[1] < 1e-5      # same K order per accumulator
    assert err(prod, w8)[1] < 1e-5
    if 2.0 * M * N * K >= 2.0e9:           # (smaller problems run the 64-row kernel whatever the switches say)
        assert not np.array_equal(prod, w8)
    if 288 < M <= 304:
        # one row block whose last row tile has at most 16 live rows: w8 multiplied it with 16 x 16 x 4 MFMAs (HALF build: 304
        # instead of 320 rows of matrix-pipe work); FC_HALF=0 is the all-32x32 build -- same K order per output
        tune("FC_HALF", "0")
        full = run()
        assert err(w8, full)[1] < 2e-6


@pytest.mark.parametrize("M,N,K,act,ldc_pad,fast", [(300, 4096, 4096, 1, 4096, True), (300, 1024, 25088, 1, 0, True),
                                                     (640, 512, 8192, 0, 0, True), (290, 520, 4096, 2, 8, False), (300, 640, 8192, 2, 4, True),
                                                     (120, 512, 2048, 1, 0, False), (300, 130, 512, 1, 6, False),
                                                     (760, 1024, 1536, 1, 0, False)])
def test_fc_pair(dev, M, N, K, act, ldc_pad, fast):
    """mnc_fc_pair: two InnerProducts of one shape in ONE launch of the 320-row LDS-DMA kernel (the box and the mask branch of a
    head stage) -- each output against torch and against its own mnc_fc call (other grouping of the K ranges: 1e-5, not bits),
    column slices of one buffer (ldc > N: Concat(fc7_mask, fc7)), the same bits on every launch; shapes that kernel would not take
    in one launch (small products, M <= 160, a ragged tail of row blocks) are exactly two mnc_fc calls."""
    rng = np.random.default_rng(M + N + K + 17)
    ld = N + ldc_pad
    ops = []
    for i in range(2):
        a = rng.normal(size=(M, K)).astype(np.float32)
        w = (rng.normal(size=(N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
        b = rng.normal(size=N).astype(np.float32)
        y = F.linear(torch.from_numpy(a), torch.from_numpy(w), torch.from_numpy(b))
        want = (F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y).numpy()
        ops.append((dev.put(a), dev.put(w), dev.put(b), want))
    # two outputs side by side in one buffer when the slice is wide enough (ldc_pad >= N), separate buffers otherwise
    if ldc_pad >= N:
        d_o0 = dev.empty((M * ld,), fill=np.nan)
        d_o1 = d_o0 + N * 4
    else:
        d_o0, d_o1 = dev.empty((M * ld,), fill=np.nan), dev.empty((M * ld,), fill=np.nan)

    def run_pair():
        dev.call("mnc_fc_pair", ops[0][0], ops[0][1], ops[0][2], d_o0, ops[1][0], ops[1][1], ops[1][2], d_o1, M, N, K, ld, act)
        full = dev.get(d_o0, (M, ld)).copy()
        o0 = full[:, :N]
        o1 = full[:, N:2 * N] if ldc_pad >= N else dev.get(d_o1, (M, ld))[:, :N].copy()
        return o0, o1

    first = run_pair()
    for got, (_, _, _, want) in zip(first, ops):
        assert not np.isnan(got).any()
        assert err(got, want)[1] < 1e-4
    for _ in range(5):
        again = run_pair()
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
    for (d_a, d_w, d_b, _), d_o in zip(ops, (d_o0, d_o1)):
        dev.call("mnc_fc", d_a, d_w, d_b, d_o, M, N, K, ld, act)
    full = dev.get(d_o0, (M, ld)).copy()
    singles = [full[:, :N], full[:, N:2 * N] if ldc_pad >= N else dev.get(d_o1, (M, ld))[:, :N].copy()]
    for got, one in zip(first, singles):
        if fast:
            assert err(got, one)[1] < 1e-5
        else:
            assert np.array_equal(got, one)
    if fast and 2.0 * M * N * K >= 8.0e9:
        assert not np.array_equal(first[0], singles[0])             # (the paired launch really cut K differently)


@pytest.mark.parametrize("M,N,K,act", FC_SHAPES + [(300, 4096, 25088, 1), (290, 512, 65536, 0), (1000, 768, 16384, 2)])
def test_fc_bf16x3(dev, M, N, K, act, tune):
    """Split-precision FC (3 bf16 MFMAs per product): measured against the float64 product.  Bar: 1e-4 of the output's
    dynamic range, the same bar as the exact-fp32 kernel (its error is ~1e-5, an fp32 GEMM's ~1e-6).  The last three shapes run
    on the 256-column LDS-DMA kernel (fc_lowp_dma_kernel<.., 0>) and are also held against the 128-column kernel (FCX3_WIDE=0)."""
    rng = np.random.default_rng(M + N + K + 1)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_wp = dev.empty(((N + 127) // 128 * 128 * K,))
    dev.call("mnc_pack_fc_bf16x3", dev.put(w), d_wp, N, K)
    d_o = dev.empty((M * N,), fill=np.nan)
    dev.call("mnc_fc_bf16x3", dev.put(a), d_wp, dev.put(b), d_o, M, N, K, N, act)
    got = dev.get(d_o, (M, N))
    y = a.astype(np.float64) @ w.astype(np.float64).T + b
    want = np.maximum(y, 0) if act == 1 else 1 / (1 + np.exp(-y)) if act == 2 else y
    assert not np.isnan(got).any()
    d, rel = err(got, want)
    print("bf16x3 M=%d N=%d K=%d: max|d|=%.3e rel=%.3e" % (M, N, K, d, rel))
    assert rel < 1e-4, (d, rel)
    if N % 256 == 0 and N >= 512:
        for force in ("0", "1"):            # the 128-column kernel, and the 256-column one also where the launcher would not pick it
            tune("FCX3_WIDE", force)
            d_alt = dev.empty((M * N,), fill=np.nan)
            dev.call("mnc_fc_bf16x3", dev.put(a), d_wp, dev.put(b), d_alt, M, N, K, N, act)
            assert err(dev.get(d_alt, (M, N)), got)[1] < 2e-6, force
        dev.tune("FCX3_WIDE", None)


@pytest.mark.parametrize("M,N,K,act", [(300, 4096, 4096, 1), (45, 150, 64, 0), (300, 256, 14 * 14 * 512, 1), (7, 4096, 25088, 1),
                                       (300, 126, 8192, 0), (1000, 512, 2048, 1), (300, 441, 256, 2),
                                       (1000, 4096, 12544, 1), (700, 300, 6272, 0), (513, 320, 4096, 1), (2000, 256, 8192, 1),
                                       (300, 4096, 25088, 1), (290, 512, 65536, 0), (1000, 768, 16384, 2), (520, 1024, 8192, 1)])
def test_fc_f16(dev, M, N, K, act, tune):
    """mnc_fc_f16: exact against torch on operands rounded to fp16 (products of halves are exact in the fp32 accumulator; only
    the summation order differs), and within fp16's 11 bits of the fp32 product.  Covers 64- / 160- / 256- / 320-row blocks,
    ragged last row blocks, N that does not fill a column tile, several K splits and a single one, and -- N a multiple of 256 with
    at least 8 stages per K split: (1000, 4096, 12544) and the last four shapes -- the 256-column LDS-DMA kernel
    (fc_f16_dma_kernel, 256- and 320-row blocks), which must also agree with the 128-column kernel it replaces (FCX3_WIDE=0)."""
    rng = np.random.default_rng(M + N + K)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_wp = dev.empty(((N + 127) // 128 * 128 * K // 2,), fill=np.nan)
    dev.call("mnc_pack_fc_f16", dev.put(w), d_wp, N, K)
    d_o = dev.empty((M * N,), fill=np.nan)
    dev.call("mnc_fc_f16", dev.put(a), d_wp, dev.put(b), d_o, M, N, K, N, act)
    got = dev.get(d_o, (M, N))

    def ref(x, y):
        z = F.linear(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(b))
        return (F.relu(z) if act == 1 else torch.sigmoid(z) if act == 2 else z).numpy()
    exact = ref(a.astype(np.float16).astype(np.float32), w.astype(np.float16).astype(np.float32))
    assert not np.isnan(got).any()
    d, rel = err(got, exact)
    d32, rel32 = err(got, ref(a, w))
    print("f16 M=%d N=%d K=%d: vs fp16-rounded operands rel=%.3e, vs fp32 rel=%.3e" % (M, N, K, rel, rel32))
    assert rel < 1e-5 and rel32 < 2e-3
    if N % 256 == 0 and N >= 512:
        tune("FCX3_WIDE", "0")
        d_old = dev.empty((M * N,), fill=np.nan)
        dev.call("mnc_fc_f16", dev.put(a), d_wp, dev.put(b), d_old, M, N, K, N, act)
        dev.tune("FCX3_WIDE", None)
        assert err(dev.get(d_old, (M, N)), got)[1] < 2e-6
    if M >= 512:        # a column slice of a wider matrix (ldc > N), as the Concat-in-place plan writes fc7 / fc7_mask
        ld = N + 64
        d_wide = dev.empty((M * ld,), fill=np.nan)
        dev.call("mnc_fc_f16", dev.put(a), d_wp, dev.put(b), d_wide, M, N, K, ld, act)
        wide = dev.get(d_wide, (M, ld))
        assert np.array_equal(wide[:, :N], got) and np.isnan(wide[:, N:]).all()


@pytest.mark.parametrize("mode", ["f16", "bf16", "bf16x3"])
@pytest.mark.parametrize("M,N,K,pre", [(300, 4096, 6272, True), (300, 1024, 4096, False), (200, 512, 2048, True), (300, 4096, 25088, True),
                                       (120, 512, 4096, False), (300, 384, 4096, False), (1000, 1024, 4096, True), (640, 512, 8192, False),
                                       (700, 512, 4096, True)])
def test_fc_lowp_pair(dev, mode, M, N, K, pre, tune):
    """mnc_fc_lowp_pair (round 6): two reduced-precision InnerProducts of one shape in one launch.  Each product against torch on the
    operands the mode rounds to (fp16 / bf16: 1e-5 of the range -- only the summation order differs; split bf16: 1e-4 against fp32),
    written as a column slice (ldc = 2 N: fc7 / fc7_mask's Concat in place), inputs as fp32 rows or stage-major, the second outputs
    bit for bit mnc_fc_pack_act of the fp32 rows; the same bits on every launch; shapes the paired kernel does not take (M <= 160,
    N % 256 != 0) equal the two single calls.  M > 320 (round 6, throughput plan): several row blocks in the one launch -- 256-row
    blocks (1000), 320-row blocks (640); 700 rows = 2 x 320 + a ragged 60-row tail stay two single calls."""
    m = {"bf16x3": 0, "f16": 1, "bf16": 2}[mode]
    rnd = (lambda x: x.astype(np.float16).astype(np.float32)) if mode == "f16" else _rbf16 if mode == "bf16" else (lambda x: x)
    rng = np.random.default_rng(M + N + K + m)
    a = [rng.normal(size=(M, K)).astype(np.float32) for _ in range(2)]
    w = [(rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32) for _ in range(2)]
    b = [rng.normal(size=N).astype(np.float32) for _ in range(2)]
    d_a = [dev.put(x) for x in a]
    d_b = [dev.put(x) for x in b]
    wpv = 1 if mode == "bf16x3" else 2                    # values per 32-bit word of the packed forms
    d_w = []
    for x in w:
        d = dev.empty(((N + 127) // 128 * 128 * K // wpv,), fill=np.nan)
        dev.call("mnc_pack_fc_" + mode, dev.put(x), d, N, K)
        d_w.append(d)
    sm_fmt = {"f16": 1, "bf16x3": 2, "bf16": 3}[mode]     # (round 6: the plain bf16 mode has its stage-major form too, format 3)
    pack_flag = {"f16": 1, "bf16x3": 0, "bf16": 2}[mode]
    use_pre = pre and sm_fmt != 0
    d_sm = [None, None]
    if use_pre:
        for i in range(2):
            d_sm[i] = dev.empty((M * K // wpv,), fill=np.nan)
            dev.call("mnc_fc_pack_act", d_a[i], d_sm[i], M, K, pack_flag)
    ld = 2 * N
    want_osm = sm_fmt != 0 and N % 64 == 0
    outs = []
    for rep in range(2):
        d_o = dev.empty((M * ld,), fill=np.nan)
        d_osm = [dev.empty((M * N // wpv,), fill=np.nan) if want_osm else None for _ in range(2)]
        dev.call("mnc_fc_lowp_pair", m, None if use_pre else d_a[0], d_sm[0], None if use_pre else d_a[1], d_sm[1], M, d_w[0], d_w[1],
                 d_b[0], d_b[1], d_o + N * 4, d_o, M, N, K, ld, 1, d_osm[0], d_osm[1], sm_fmt if want_osm else 0)
        o = dev.get(d_o, (M, ld))
        outs.append((o, [dev.get(x, (M * N // wpv,)).view(np.uint32) if x else None for x in d_osm]))
    o = outs[0][0]
    assert np.array_equal(o, outs[1][0]) and not np.isnan(o).any()
    got = [o[:, N:], o[:, :N]]
    for i in range(2):
        ref = np.maximum(rnd(a[i]).astype(np.float64) @ rnd(w[i]).T.astype(np.float64) + b[i], 0)
        d, rel = err(got[i], ref)
        assert rel < (1e-4 if mode == "bf16x3" else 1e-5), (i, d, rel)
        if want_osm:
            d_chk = dev.empty((M * N // wpv,), fill=np.nan)
            dev.call("mnc_fc_pack_act", dev.put(np.ascontiguousarray(got[i])), d_chk, M, N, pack_flag)
            assert np.array_equal(dev.get(d_chk, (M * N // wpv,)).view(np.uint32), outs[0][1][i])
    if M <= 160 or N % 256 or M == 700:       # not paired: the two single calls, bit for bit
        for i in range(2):
            d_s = dev.empty((M * N,), fill=np.nan)
            dev.call("mnc_fc_" + mode, d_a[i], d_w[i], d_b[i], d_s, M, N, K, N, 1)
            assert np.array_equal(dev.get(d_s, (M, N)), got[i])
    if K >= 16384 and M > 160 and N % 256 == 0:
        # PLAN (round 6): the default cuts K into ranges of >= 2048 values on at most half the chip; PLAN=1 is rounds 3-5's full cut
        # (= FC_SPLIT_DIV=0): the same product, another grouping of the partial sums
        def again():
            d_o2 = dev.empty((M * ld,), fill=np.nan)
            dev.call("mnc_fc_lowp_pair", m, None if use_pre else d_a[0], d_sm[0], None if use_pre else d_a[1], d_sm[1], M, d_w[0], d_w[1],
                     d_b[0], d_b[1], d_o2 + N * 4, d_o2, M, N, K, ld, 1, None, None, 0)
            return dev.get(d_o2, (M, ld))
        tune("PLAN", "1")
        lat = again()
        dev.tune("PLAN", None)
        tune("FC_SPLIT_DIV", "0")
        full = again()
        dev.tune("FC_SPLIT_DIV", None)
        assert np.array_equal(lat, full) and not np.array_equal(lat, o)
        assert err(lat, o)[1] < (1e-4 if mode == "bf16x3" else 1e-5)


def test_fc_column_slice_and_pack(dev):
    """ldc > N writes a column slice (Concat in place); mnc_pack_fc_weights permutes (c,h,w) columns to (h,w,c)."""
    rng = np.random.default_rng(9)
    R, C, P, N = 20, 32, 7, 96
    feat = rng.normal(size=(R, C, P, P)).astype(np.float32)
    w = (rng.normal(size=(N, C * P * P)) * 0.02).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    d_wp = dev.empty((w.size,))
    dev.call("mnc_pack_fc_weights", dev.put(w), d_wp, N, C, P, P)
    wp = dev.get(d_wp, (N, P * P, C))
    assert np.array_equal(wp, w.reshape(N, C, P * P).transpose(0, 2, 1))
    wide = dev.empty((R * 2 * N,), fill=7.0)
    dev.call("mnc_fc", dev.put(np.ascontiguousarray(feat.transpose(0, 2, 3, 1))), d_wp, dev.put(b), wide + N * 4, R, N,
             C * P * P, 2 * N, 0)
    out = dev.get(wide, (R, 2 * N))
    want = feat.reshape(R, -1) @ w.T + b
    assert np.all(out[:, :N] == 7.0)
    assert err(out[:, N:], want)[1] < 1e-4


def test_softmax_rows_and_eltwise(dev):
    x = np.random.default_rng(10).normal(size=(300, 21)).astype(np.float32) * 3
    d_o = dev.empty((x.size,))
    dev.call("mnc_softmax_rows", dev.put(x), d_o, 300, 21)
    assert err(dev.get(d_o, x.shape), F.softmax(torch.from_numpy(x), dim=1).numpy())[0] < 1e-6
    wide = np.random.default_rng(11).normal(size=(300, 126)).astype(np.float32)
    dev.call("mnc_softmax_rows_ld", dev.put(wide) + 21 * 4, 126, d_o, 300, 21)
    assert err(dev.get(d_o, x.shape), F.softmax(torch.from_numpy(wide[:, 21:42].copy()), dim=1).numpy())[0] < 1e-6
    dev.call("mnc_eltwise", dev.put(x), d_o, x.size, 2)
    assert err(dev.get(d_o, x.shape), torch.sigmoid(torch.from_numpy(x)).numpy())[0] < 1e-6
    dev.call("mnc_eltwise", dev.put(x), d_o, x.size, 1)
    assert np.array_equal(dev.get(d_o, x.shape), np.maximum(x, 0))


def test_argument_errors_are_reported(dev):
    from mnc_amd import _lib
    with pytest.raises(_lib.MncError) as e:
        dev.call("mnc_conv3x3", 0, 0, 0, 0, 8, 8, 8, 32, 1)
    assert e.value.code == 1 and "null" in str(e.value)
    p = dev.empty((64,))
    with pytest.raises(_lib.MncError):
        dev.call("mnc_conv3x3", p, p, p, p, 8, 8, 12, 32, 1)        # Cin % 8 != 0
    with pytest.raises(_lib.MncError):
        dev.call("mnc_fc", p, p, p, p, 4, 4, 33, 4, 0)              # K % 32 != 0


def test_profiling_records(dev):
    x = np.zeros((8, 16, 16), np.float32)
    d_y = dev.empty((8 * 8 * 8,))
    dev.call("mnc_prof_enable", 1)
    dev.call("mnc_prof_reset")
    for _ in range(3):
        dev.call("mnc_maxpool2_c8", dev.put(to_c8(x)), d_y, 8, 16, 16)
    import ctypes
    n = ctypes.c_int(0)
    dev.call("mnc_prof_count", ctypes.addressof(n))
    assert n.value == 3
    name = ctypes.create_string_buffer(64)
    ms = ctypes.c_float(-1)
    dev.call("mnc_prof_get", 0, ctypes.addressof(name), 64, ctypes.addressof(ms), None, None)
    assert name.value == b"maxpool2_c8" and ms.value >= 0
    dev.call("mnc_prof_enable", 0)
    dev.call("mnc_prof_reset")


# ---- size-independent properties at BASELINE's full sizes (the oracle is too slow there) -----------------------------------
@pytest.mark.parametrize("fn,pack,pitch", [("mnc_conv3x3", "mnc_pack_conv3x3_weights", 76),
                                           ("mnc_conv3x3_bf16x3", "mnc_pack_conv3x3_bf16x3", 84),
                                           ("mnc_conv3x3_f16", "mnc_pack_conv3x3_f16", 84), ("mnc_conv3x3_bf16", "mnc_pack_conv3x3_bf16", 84)])
def test_conv3x3_full_size_linearity_and_shift(dev, fn, pack, pitch):
    """conv3_2's shape (150x250, 256 -> 256), no bias / ReLU: scaling the input by a power of two scales the output exactly
    (fp32, split-bf16, fp16 and bf16 arithmetic alike: rounding commutes with a power of two), an impulse input reproduces the
    (flipped) filter taps to the mode's precision, and a zero input gives exactly the bias."""
    H, W, Cin, Cout = 150, 250, 256, 256
    rng = np.random.default_rng(5)
    x = rng.normal(size=(Cin, H, W)).astype(np.float32)
    w = (rng.normal(size=(Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b0, b1 = np.zeros(Cout, np.float32), rng.normal(size=Cout).astype(np.float32)
    d_w = dev.empty(((Cin // 8) * Cout * pitch,))
    dev.call(pack, dev.put(w), d_w, Cout, Cin)

    def run(inp, bias):
        d_y = dev.empty((Cout * H * W,), fill=np.nan)
        dev.call(fn, dev.put(to_c8(inp)), d_w, dev.put(bias), d_y, H, W, Cin, Cout, 0)
        return from_c8(dev.get(d_y, (Cout * H * W,)), Cout, H, W)

    y = run(x, b0)
    assert np.isfinite(y).all()
    if fn != "mnc_conv3x3_f16":          # (fp16: inputs below 2^-14 are subnormal -- x and 4 x do not round alike)
        assert np.array_equal(run(x * np.float32(4.0), b0), y * np.float32(4.0))
    z = run(np.zeros_like(x), b1)
    assert np.array_equal(z, np.broadcast_to(b1[:, None, None], z.shape))
    imp = np.zeros_like(x)
    imp[17, 70, 123] = 1.0
    r = run(imp, b0)
    tol = {"mnc_conv3x3": 0, "mnc_conv3x3_bf16x3": 2.0 ** -15, "mnc_conv3x3_f16": 2.0 ** -11, "mnc_conv3x3_bf16": 2.0 ** -8}[fn]
    for kh in range(3):
        for kw in range(3):
            got, want = r[:, 70 + 1 - kh, 123 + 1 - kw], w[:, 17, kh, kw]
            assert np.all(np.abs(got - want) <= tol * np.abs(want) + (2.0 ** -25 if fn == "mnc_conv3x3_f16" else 0.0)), (kh, kw)   # (fp16 subnormals)
    r[:, 69:72, 122:125] = 0
    assert not r.any()


@pytest.mark.parametrize("fn,pack", [("mnc_fc", None), ("mnc_fc_bf16x3", "mnc_pack_fc_bf16x3")])
def test_fc_full_size_linearity(dev, fn, pack):
    """fc6's shape (300 x 25088 -> 4096), no bias / activation: exact power-of-two scaling, zero input -> bias, and a one-hot
    row selects a weight column (exactly in fp32, to 2^-15 in split bf16)."""
    M, N, K = 300, 4096, 25088
    rng = np.random.default_rng(6)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    b0, b1 = np.zeros(N, np.float32), rng.normal(size=N).astype(np.float32)
    d_w = dev.put(w)
    if pack:
        d_wp = dev.empty(((N + 127) // 128 * 128 * K,))
        dev.call(pack, d_w, d_wp, N, K)
        d_w = d_wp

    def run(inp, bias):
        d_o = dev.empty((M * N,), fill=np.nan)
        dev.call(fn, dev.put(inp), d_w, dev.put(bias), d_o, M, N, K, N, 0)
        return dev.get(d_o, (M, N))

    y = run(a, b0)
    assert np.isfinite(y).all()
    assert np.array_equal(run(a * np.float32(0.5), b0), y * np.float32(0.5))
    assert np.array_equal(run(np.zeros_like(a), b1), np.broadcast_to(b1, (M, N)))
    onehot = np.zeros_like(a)
    cols = rng.integers(0, K, M)
    onehot[np.arange(M), cols] = 1.0
    got, want = run(onehot, b0), w[:, cols].T
    tol = 0 if fn == "mnc_fc" else 2.0 ** -15
    assert np.all(np.abs(got - want) <= tol * np.abs(want))


@pytest.mark.parametrize("fh,fw,seed,quant", [(38, 63, 3, False), (12, 20, 2, False), (38, 63, 5, True), (63, 63, 6, False),
                                              (50, 84, 7, True)])
def test_proposal_layer_golden_and_both_topk_paths(dev, golden, monkeypatch, fh, fw, seed, quant, tune):
    """mnc_proposal (device-resident ProposalLayer.forward, lib/pylayer/proposal_layer.py:52-175) on stand-alone RPN outputs:
    rois == the oracle's ProposalLayer == the reference's own layer (golden `prop_*_rois`), bit for bit, with the multi-workgroup
    top-K (sorted runs + rank merge) and with the single-workgroup radix select (MNC_TOPK_SINGLE_WG=1); scores quantised to 1/64
    give thousands of exact ties (order: score descending, anchor index ascending) and boxes below the minimum size are filtered
    -- both top-K paths must agree on every candidate."""
    import golden_inputs as GI
    from oracle import host as ohost
    from transform.anchors import generate_anchors
    pc = GI.proposal_case(fh, fw, seed)
    prob, bbox, info = pc["cls_prob"].copy(), pc["bbox_pred"].copy(), pc["im_info"]
    if quant:
        prob = (np.round(prob * 64) / 64).astype(np.float32)
        bbox[:, 2::4] -= 2.5                        # many boxes shrink below RPN_MIN_SIZE and are filtered
    anchors = np.ascontiguousarray(generate_anchors(), np.float32)
    d_prob, d_bbox = dev.put(prob), dev.put(bbox)
    d_rois = dev.empty((300, 5))
    want = ohost.proposal_forward(prob, bbox, info)
    results = []
    for single in (False, True):
        if single:
            tune("TOPK_SINGLE_WG", "1")
        else:
            dev.tune("TOPK_SINGLE_WG", None)
        num = ctypes.c_int(-1)
        dev.call("mnc_proposal", d_prob, d_bbox, 9, fh, fw, _lib.ptr(anchors), 16, float(info[0, 0]), float(info[0, 1]),
                 float(info[0, 2]), 6000, 300, 0.7, 16.0, d_rois, ctypes.addressof(num))
        rois = dev.get(d_rois, (300, 5))[:num.value]
        n = ctypes.c_int(0)
        dev.call("mnc_proposal_candidates", None, None, 0, ctypes.addressof(n))
        cb, cs = np.zeros((n.value, 4), np.float32), np.zeros(n.value, np.float32)
        if n.value:
            dev.call("mnc_proposal_candidates", _lib.ptr(cb), _lib.ptr(cs), n.value, ctypes.addressof(n))
        results.append((rois, cb, cs))
        assert rois.shape == want.shape and np.array_equal(rois, want), ("single" if single else "wide")
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])
    ob, osc = ohost.proposal_candidates(prob, bbox, info)
    assert np.array_equal(results[0][2], osc.ravel()) and np.array_equal(results[0][1], ob)
    if not quant and (fh, fw, seed) in ((38, 63, 3), (12, 20, 2)):
        tag = "full" if fh == 38 else "small"
        assert np.array_equal(want, golden["prop_%s_rois" % tag])
