"""The SPEC-CHOICEs of ROIWarping / MaskResize / MaskPooling as run-time configuration (include/mnc_hip.h: mnc_layer_conventions;
oracle/SPEC.md section 6).  The layers' source (caffe-mnc) is not available -- parity of these three layers is UNPINNED whichever
convention is selected -- so what is tested is that every alternative is reachable without recompiling and computes exactly what
it says on every level:

  CPU   oracle C (orc_*_ex) == the numpy statement of each alternative (oracle/spec_alternatives.py);
        caffe.Net(layer_conventions=...) / cfg.LAYER_CONVENTIONS reach the context (fake backend).
  GPU   HIP kernels == oracle C, bit for bit, per alternative and for combinations, all launch variants (plain / fused pool /
        second outputs); the engine and the one-call native pipeline under a non-default combination == the oracle heads under
        the same combination (teacher-forced protocol of tests/test_gpu_engine.py)."""
import numpy as np
import pytest

import golden_inputs as GI
import mnc_amd
from oracle import native
from oracle import spec_alternatives as sa

mnc_amd.install_paths()

# (name, conventions for the library / oracle C, kwargs of the numpy alternative: roi_warp, mask_resize, mask_pool)
ALTERNATIVES = [
    ("spec", {}, {}, {}, {}),
    ("warp_center", {"warp_sample": 1}, {"sample": "center"}, {}, {}),
    ("warp_center_half", {"warp_sample": 2}, {"sample": "center_half"}, {}, {}),
    ("warp_round_edges", {"warp_round_edges": 1}, {"round_edges": True}, {}, {}),
    ("warp_no_plus_one", {"warp_no_plus_one": 1}, {"plus_one": False}, {}, {}),
    ("warp_clamp", {"warp_oob": 1}, {"oob": "clamp"}, {}, {}),
    ("resize_half_pixel", {"resize_mode": 1}, {}, {"mode": "half_pixel"}, {}),
    ("resize_align_corners", {"resize_mode": 2}, {}, {"mode": "align_corners"}, {}),
    ("maskpool_binary", {"maskpool_binary": 1, "maskpool_thresh": 0.4}, {}, {}, {"binary": True}),
    ("combination", {"warp_sample": 2, "warp_oob": 1, "warp_no_plus_one": 1, "resize_mode": 1, "maskpool_binary": 1},
     {"sample": "center_half", "oob": "clamp", "plus_one": False}, {"mode": "half_pixel"}, {"binary": True}),
]
IDS = [a[0] for a in ALTERNATIVES]


def _rois(rng, R, W, H):
    b = GI._boxes(rng, R, W, H, 8, 500)
    b[0] = [0, 0, W - 1, H - 1]                  # whole image
    b[1] = [10.3, 20.7, 10.9, 21.2]              # sub-pixel
    b[2] = [W - 1, H - 1, W - 1, H - 1]          # bottom-right corner
    b[3] = [200, 100, 150, 60]                   # malformed x2 < x1
    b[4] = [-40, -25, 30, 20]                    # partly outside the map
    b[5] = [W + 40, H + 40, W + 90, H + 70]      # wholly outside
    return np.hstack([np.zeros((R, 1), np.float32), b]).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("name,conv,wk,rk,pk", ALTERNATIVES, ids=IDS)
def test_oracle_switches_equal_the_numpy_alternatives(name, conv, wk, rk, pk):
    rng = np.random.default_rng(3)
    feat = rng.normal(size=(16, 38, 63)).astype(np.float32)
    rois = _rois(rng, 24, 1000, 600)
    masks = rng.uniform(0, 1, (24, 1, 21, 21)).astype(np.float32)
    f14 = rng.normal(size=(24, 16, 14, 14)).astype(np.float32)
    with native.conventions(**conv):
        for P in (7, 14, 28):
            assert np.array_equal(native.roi_warp(feat, rois, P, P, 0.0625), sa.roi_warp(feat, rois, P, P, 0.0625, **wk)), P
        m14 = native.mask_resize(masks, 14, 14)
        assert float(np.abs(m14 - sa.mask_resize(masks, 14, 14, **rk)).max()) < 5e-7     # separable vs four-product form
        assert np.array_equal(native.mask_pool(f14, m14), sa.mask_pool(f14, m14, **pk))
    # and the switches are off again afterwards
    assert np.array_equal(native.roi_warp(feat, rois, 14, 14, 0.0625), sa.roi_warp(feat, rois, 14, 14, 0.0625))


def test_unknown_or_invalid_conventions_are_rejected():
    from mnc_amd.native_net import LayerConventions
    with pytest.raises(KeyError):
        LayerConventions.make({"warp_smaple": 1})
    with pytest.raises(KeyError):
        native.conventions(resize="half")
    c = LayerConventions.make({"warp_sample": 2, "maskpool_thresh": 0.5})
    assert c.as_dict() == {"warp_sample": 2, "warp_round_edges": 0, "warp_no_plus_one": 0, "warp_oob": 0, "resize_mode": 0,
                           "maskpool_binary": 0, "maskpool_thresh": 0.5}


def test_net_config_struct_matches_the_header():
    """mnc_net_config ends with the conventions struct; sizes agree with what a C compiler makes of include/mnc_hip.h."""
    import ctypes
    import os
    import subprocess
    import tempfile
    from mnc_amd.native_net import LayerConventions, NetConfig
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "mnc_hip.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(mnc_net_config), '
           'sizeof(mnc_layer_conventions), offsetof(mnc_net_config, conventions));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(repo, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        a, b, c = map(int, subprocess.check_output([os.path.join(d, "s")]).split())
    assert (a, b, c) == (ctypes.sizeof(NetConfig), ctypes.sizeof(LayerConventions), NetConfig.conventions.offset)


def test_engine_net_routes_conventions_to_the_context(monkeypatch):
    """caffe.Net(..., layer_conventions=...) and cfg.LAYER_CONVENTIONS -> mnc_ctx_set_layer_conventions; with the fake backend (its
    RoI entry points are the oracle's) a forward under a convention equals the oracle head under the same convention."""
    import fake_backend
    from mnc_amd import models, synth
    from mnc_config import cfg
    from oracle import net as onet
    fake_backend.install(monkeypatch)
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    conv = {"warp_sample": 1, "resize_mode": 2, "maskpool_binary": 1}
    data = np.random.default_rng(0).uniform(-120, 130, (1, 3, 64, 96)).astype(np.float32)
    im_info = np.array([[64, 96, 1.0]], np.float32)
    try:
        for how in ("kwarg", "cfg"):
            if how == "kwarg":
                net = Net(path, w, 1, layer_conventions=conv)
            else:
                monkeypatch.setitem(cfg, "LAYER_CONVENTIONS", conv)
                net = Net(path, w, 1)
            assert net.layer_conventions.as_dict()["resize_mode"] == 2
            net.blobs["data"].reshape(*data.shape)
            net.forward(data=data, im_info=im_info)
            rois = net.blobs["rois"].data.copy()
            c5 = net.blobs["conv5_3"].data.copy()
            with native.conventions(**conv):
                want = onet.head(w, c5, rois, False)
            with native.conventions(**native.SPEC_CONVENTIONS):     # (the fake's context IS the oracle's global switches)
                spec = onet.head(w, c5, rois, False)
            got = net.blobs["mask_proposal"].data
            assert float(np.abs(got - want["mask_proposal"]).max()) < 1e-4
            assert float(np.abs(got - spec["mask_proposal"]).max()) > 1e-3          # the convention really changed the result
            net.close()
    finally:
        native._conv.clear()
        native._conv.update(native.SPEC_CONVENTIONS)


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    from gpu_util import Dev
    d = Dev(0)
    yield d
    d.close()


def _set(dev, conv):
    import ctypes
    from mnc_amd.native_net import LayerConventions
    c = LayerConventions.make(conv)
    dev.call("mnc_ctx_set_layer_conventions", ctypes.addressof(c))
    back = LayerConventions()
    dev.call("mnc_ctx_get_layer_conventions", ctypes.addressof(back))
    assert back.as_dict() == c.as_dict()


@pytest.mark.gpu
@pytest.mark.parametrize("name,conv,wk,rk,pk", ALTERNATIVES, ids=IDS)
def test_hip_kernels_follow_every_convention(dev, name, conv, wk, rk, pk):
    """mnc_roi_warp (14x14 direct, 28x28 + fused MAX pool, 7x7), with and without the InnerProduct's second output;
    mnc_mask_resize; mnc_mask_pool (plain, fused pool, second output); mnc_box_mask_pool -- bit for bit the oracle C under the
    same switches (which is bit for bit / 5e-7 the numpy alternative, CPU test above)."""
    from gpu_util import to_c8
    rng = np.random.default_rng(11)
    C, H, W, R = 64, 38, 63, 40
    feat = rng.normal(size=(C, H, W)).astype(np.float32)
    rois = _rois(rng, R, 1000, 600)
    masks = rng.uniform(0, 1, (R, 1, 21, 21)).astype(np.float32)
    d_feat, d_rois = dev.put(to_c8(feat)), dev.put(rois)
    _set(dev, conv)
    try:
        with native.conventions(**conv):
            for pool2, P in ((0, 14), (1, 14), (0, 7)):
                want = native.maxpool2(native.roi_warp(feat, rois, 2 * P, 2 * P, 0.0625)) if pool2 else \
                    native.roi_warp(feat, rois, P, P, 0.0625)
                for fmt in (0, 1, 2):
                    d_out = dev.empty((R * P * P * C,), fill=np.nan)
                    d_sm = dev.empty((R * P * P * C,), fill=0) if fmt else None
                    dev.call("mnc_roi_warp_sm", d_feat, C, H, W, d_rois, R, P, P, 0.0625, pool2, d_out, d_sm, fmt)
                    got = dev.get(d_out, (R, P, P, C)).transpose(0, 3, 1, 2)
                    assert np.array_equal(got, want), (pool2, P, fmt)
                if wk:
                    assert not np.array_equal(want, native.maxpool2(sa.roi_warp(feat, rois, 2 * P, 2 * P, 0.0625)) if pool2
                                              else sa.roi_warp(feat, rois, P, P, 0.0625))
            # MaskResize
            d_m14 = dev.empty((R * 196,), fill=np.nan)
            dev.call("mnc_mask_resize", dev.put(masks), d_m14, R, 21, 21, 14, 14)
            m14 = dev.get(d_m14, (R, 1, 14, 14))
            assert np.array_equal(m14, native.mask_resize(masks, 14, 14))
            # MaskPooling, MaskPooling + pool, box + mask pools in one pass
            f14 = rng.normal(size=(R, C, 14, 14)).astype(np.float32)
            d_f14 = dev.put(np.ascontiguousarray(f14.transpose(0, 2, 3, 1)))
            want_mp = native.mask_pool(f14, m14)
            for fmt in (0, 1, 2):
                d_o = dev.empty((R * 196 * C,), fill=np.nan)
                d_sm = dev.empty((R * 196 * C,), fill=0) if fmt else None
                dev.call("mnc_mask_pool_sm", d_f14, d_m14, d_o, R, 14, 14, C, 0, d_sm, fmt)
                assert np.array_equal(dev.get(d_o, (R, 14, 14, C)).transpose(0, 3, 1, 2), want_mp), fmt
                d_o2 = dev.empty((R * 49 * C,), fill=np.nan)
                dev.call("mnc_mask_pool_sm", d_f14, d_m14, d_o2, R, 14, 14, C, 1, d_sm, fmt)
                assert np.array_equal(dev.get(d_o2, (R, 7, 7, C)).transpose(0, 3, 1, 2), native.maxpool2(want_mp)), fmt
                d_b, d_m = dev.empty((R * 49 * C,), fill=np.nan), dev.empty((R * 49 * C,), fill=np.nan)
                d_sb = dev.empty((R * 49 * C,), fill=0) if fmt else None
                d_sk = dev.empty((R * 49 * C,), fill=0) if fmt else None
                dev.call("mnc_box_mask_pool", d_f14, d_m14, d_b, d_m, R, 14, 14, C, d_sb, d_sk, fmt)
                assert np.array_equal(dev.get(d_b, (R, 7, 7, C)).transpose(0, 3, 1, 2), native.maxpool2(f14))
                assert np.array_equal(dev.get(d_m, (R, 7, 7, C)).transpose(0, 3, 1, 2), native.maxpool2(want_mp)), fmt
    finally:
        _set(dev, {})


@pytest.mark.gpu
def test_invalid_conventions_fail_loudly(dev):
    from mnc_amd import _lib
    with pytest.raises(_lib.MncError, match="warp_sample"):
        _set(dev, {"warp_sample": 3})
    with pytest.raises(_lib.MncError, match="resize_mode"):
        _set(dev, {"resize_mode": -1})
    _set(dev, {})


@pytest.mark.gpu
def test_net_config_inherits_or_applies_conventions(dev):
    """mnc_net_create and the conventions in force on its context: the default config (conventions.inherit = 1) leaves them alone;
    a config with inherit = 0 applies its member -- an all-zero one too, i.e. a host CAN ask for the SPEC on a context an earlier
    net or call had changed (the all-zero value used to mean "inherit", so the result depended on creation order)."""
    import ctypes
    from mnc_amd import _lib
    from mnc_amd.native_net import LayerConventions, default_config

    def in_force():
        back = LayerConventions()
        dev.call("mnc_ctx_get_layer_conventions", ctypes.addressof(back))
        return back.as_dict()

    def create(cfg):
        h = ctypes.c_void_p()
        _lib.call("mnc_net_create", dev.h, ctypes.addressof(cfg), ctypes.addressof(h))
        _lib.call("mnc_net_destroy", h.value)

    alt = {"warp_sample": 2, "resize_mode": 1}
    try:
        _set(dev, alt)
        want = in_force()
        cfg = default_config()
        assert cfg.conventions.inherit == 1
        create(cfg)
        assert in_force() == want                                   # inherited
        cfg.conventions = LayerConventions.make({})                 # explicit SPEC (inherit = 0)
        create(cfg)
        assert in_force() == LayerConventions.make({}).as_dict()
        cfg.conventions = LayerConventions.make({"maskpool_binary": 1})
        create(cfg)
        assert in_force()["maskpool_binary"] == 1
    finally:
        _set(dev, {})


@pytest.mark.gpu
@pytest.mark.parametrize("conv", [{"warp_sample": 2, "warp_oob": 1, "resize_mode": 1},
                                  {"warp_round_edges": 1, "warp_no_plus_one": 1, "resize_mode": 2, "maskpool_binary": 1}],
                         ids=["roialign_like", "roipool_like"])
def test_engine_and_native_pipeline_under_alternative_conventions(conv):
    """Whole reduced-width graph under a non-default combination: the Python engine against the oracle heads evaluated under the
    same switches (teacher-forced, tests/test_gpu_engine.py:check_forward), the one-call native pipeline (mnc_net_config::
    conventions) bit-identical to the engine, and both different from the SPEC's result."""
    from mnc_amd import models, synth
    from mnc_amd.engine import Net
    from mnc_amd.native_net import NativeNet
    from test_gpu_engine import check_forward
    from test_gpu_pipeline import _check_against_engine
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    net = Net(path, w, 1, layer_conventions=conv)
    spec_net = Net(path, w, 1)
    nat = NativeNet(w, layer_conventions=conv)
    try:
        rng = np.random.default_rng(5)
        data = rng.uniform(-120, 130, (1, 3, 96, 160)).astype(np.float32)
        im_info = np.array([[96, 160, 1.0]], np.float32)
        for n in (net, spec_net):
            n.blobs["data"].reshape(*data.shape)
            n.forward(data=data, im_info=im_info)
        with native.conventions(**conv):
            check_forward(net, w, data, im_info)
        a, b = net.blobs["mask_proposal"]._host_read(), spec_net.blobs["mask_proposal"]._host_read()
        assert np.array_equal(net.blobs["rois"]._host_read(), spec_net.blobs["rois"]._host_read())
        assert float(np.abs(a - b).max()) > 1e-3
        for _ in range(3):                                  # eager, graph capture, graph replay
            im = rng.integers(0, 256, (75, 100, 3), dtype=np.uint8)
            _check_against_engine(nat, net, im)
    finally:
        nat.close()
        net.close()
        spec_net.close()
