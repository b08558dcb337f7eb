"""GPU: NMS and mask voting through the C ABI (mnc_nms / mnc_mv and the nms.gpu_nms / nms.mv wrappers) against
  (1) the fixtures produced by the reference's own code (tests/golden), (2) the CPU oracle on the same seeded inputs,
  (3) the reference's own .cu code compiled for the CPU (oracle/_ref, prebuilt) when present.
Bit-exact everywhere: keep indices, bitmask words, int boxes, float32 masks."""
import ctypes

import numpy as np
import pytest

import golden_inputs as GI
import mnc_amd
from mnc_amd import _lib
from oracle import native

pytestmark = pytest.mark.gpu
mnc_amd.install_paths()


def _sorted(dets):
    order = dets[:, 4].argsort()[::-1]
    return np.ascontiguousarray(dets[order]), order


@pytest.mark.parametrize("n,thr,seed", GI.NMS_CASES)
def test_nms_keep_vs_reference_fixture(golden, n, thr, seed):
    from nms.gpu_nms import gpu_nms
    dets = GI.nms_case(n, seed)
    keep = gpu_nms(dets, thr, 0)
    want = golden["nms_%d_%s_keep" % (n, str(thr).replace(".", "p"))]
    assert np.array_equal(np.array(keep, np.int64), want)
    assert keep == native.gpu_nms(dets, thr)


@pytest.mark.parametrize("n,thr,seed", [(6000, 0.7, 31), (600, 0.3, 32), (129, 0.5, 33), (64, 0.5, 34), (1, 0.5, 35)])
def test_nms_bitmask_words(n, thr, seed):
    boxes, _ = _sorted(GI.nms_case(n, seed))
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), np.uint64)
    _lib.call("mnc_nms_mask", _lib.ptr(mask), _lib.ptr(boxes), n, 5, float(thr), 0)
    want = native.nms_mask(boxes, thr)
    rt = np.arange(n)[:, None] // 64
    upper = np.arange(cb)[None, :] >= rt                      # the scan only reads column tiles >= the row tile
    assert np.array_equal(mask[upper], want[upper])
    assert not mask[~upper].any()


def test_reference_cxx_binding_runs(golden, tmp_path):
    """b1/b2 as the reference binds them: a C++ program that sees `_nms` / `_mv` only through the reference's headers
    (tests/c/ref_binding_main.cpp), linked to the mangled exports of libmnc_hip.so, fed the fixture inputs through files --
    keep lists and voted masks / boxes equal the reference-generated fixtures."""
    import subprocess
    from test_abi_cpu import build_ref_binding_program
    exe = build_ref_binding_program(str(tmp_path / "ref_binding_main"))
    for n, thr, seed in GI.NMS_CASES:
        dets = GI.nms_case(n, seed)
        srt, order = _sorted(dets)
        srt.tofile(str(tmp_path / "dets.f32"))
        subprocess.run([exe, "nms", str(tmp_path / "dets.f32"), str(n), repr(float(thr)), str(tmp_path / "keep.i32")], check=True,
                       timeout=300)
        out = np.fromfile(str(tmp_path / "keep.i32"), np.int32)
        assert out[0] == len(out) - 1
        want = golden["nms_%d_%s_keep" % (n, str(thr).replace(".", "p"))]
        assert np.array_equal(order[out[1:]], want)             # gpu_nms.pyx:31 maps sorted positions back through `order`
    mc = GI.mv_case(8)
    d = tmp_path / "mv"
    d.mkdir()
    boxes = np.ascontiguousarray(mc["boxes"][:, :4], np.float32)
    for name, a, t in (("boxes.f32", boxes, np.float32), ("masks.f32", mc["masks"], np.float32), ("wts.f32", mc["weights"], np.float32),
                       ("inds.i32", mc["inds"], np.int32), ("start.i32", mc["start"], np.int32)):
        np.ascontiguousarray(a, t).tofile(str(d / name))
    S = mc["masks"].shape[-1]
    R = len(mc["start"])
    subprocess.run([exe, "mv", str(d), str(len(boxes)), str(len(mc["inds"])), str(R), str(mc["H"]), str(mc["W"]), str(S)], check=True,
                   timeout=300)
    assert np.array_equal(np.fromfile(str(d / "out_box.i32"), np.int32).reshape(R, 4), golden["mv_box"])
    assert np.array_equal(np.fromfile(str(d / "out_mask.f32"), np.float32).reshape(golden["mv_mask"].shape), golden["mv_mask"])


def test_reference_cython_extensions_on_libmnc_hip(golden):
    """The reference's OWN extension modules -- lib/nms/gpu_nms.pyx and gpu_mv.pyx cythonized as C++ against the reference's
    gpu_nms.hpp / gpu_mv.hpp and linked with -lmnc_hip instead of the two .cu files (oracle/build_ref_ext.py; INTEGRATION.md A,
    executed where /root/reference is mounted, prebuilt modules travel here) -- called on the fixture inputs: the reference's
    Python-visible results, produced by the HIP kernels."""
    import importlib
    import os
    import sys
    ext = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ext")
    if not (os.path.isfile(os.path.join(ext, "gpu_nms.so")) and os.path.isfile(os.path.join(ext, "gpu_mv.so"))):
        pytest.skip("oracle/_ref/ext not built (needs /root/reference + Cython at build time)")
    sys.path.insert(0, ext)
    try:
        ref_nms = importlib.import_module("gpu_nms")
        ref_mv = importlib.import_module("gpu_mv")
    finally:
        sys.path.remove(ext)
    for n, thr, seed in GI.NMS_CASES:
        keep = ref_nms.gpu_nms(GI.nms_case(n, seed), float(thr), 0)
        assert np.array_equal(np.array(keep, np.int64), golden["nms_%d_%s_keep" % (n, str(thr).replace(".", "p"))])
    mc = GI.mv_case(8)
    rm, rb = ref_mv.mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"], 0)
    assert np.array_equal(rb, golden["mv_box"]) and np.array_equal(rm, golden["mv_mask"])


def test_nms_topk_prefix_and_edges():
    dets, _ = _sorted(GI.nms_case(6000, 41))
    keep = np.zeros(6000, np.int32)
    num = ctypes.c_int(0)
    _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 6000, 5, 0.7, 0)
    full = keep[:num.value].copy()
    assert np.array_equal(full, native.nms_sorted(dets, 0.7))
    for k in (1, 64, 300, 301):
        _lib.call("mnc_nms_topk", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 6000, 5, 0.7, k, 0)
        assert num.value == min(k, len(full)) and np.array_equal(keep[:num.value], full[:k])
    # n == 0 is legal; a bad device id is an error with a message, not a crash
    _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), None, 0, 5, 0.7, 0)
    assert num.value == 0
    with pytest.raises(_lib.MncError) as e:
        _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 10, 5, 0.7, 99)
    assert "device" in str(e.value)
    # the void wrapper with the reference's exact signature
    lib = _lib.load()
    lib._nms(_lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 600, 5, ctypes.c_float(0.7), 0)
    assert np.array_equal(keep[:num.value], native.nms_sorted(dets[:600], 0.7))


def test_nms_batched_equals_per_class_calls():
    """mnc_nms_batched (one launch pair for all classes) == 20 independent gpu_nms calls == oracle, incl. max_keep."""
    from nms.gpu_nms import gpu_nms, gpu_nms_batched
    vc = GI.voting_case(600, 600, 1000, 21)
    boxes, scores = vc["boxes"], vc["scores"]
    full = gpu_nms_batched(boxes, scores[:, 1:], 0.3, 0)
    top = gpu_nms_batched(boxes, scores[:, 1:], 0.3, 0, max_keep=100)
    for c in range(1, 21):
        dets = np.hstack((boxes, scores[:, c:c + 1]))
        want = native.gpu_nms(dets, 0.3)
        assert full[c - 1] == want == gpu_nms(dets, 0.3, 0)
        assert top[c - 1] == want[:100]
    big = GI.nms_case(6000, 55)
    sc = np.stack([big[:, 4], big[::-1, 4].copy(), np.roll(big[:, 4], 17)], 1)
    got = gpu_nms_batched(big[:, :4], sc, 0.7, 0, max_keep=300)
    for b in range(3):
        assert got[b] == native.gpu_nms(np.hstack((big[:, :4], sc[:, b:b + 1])), 0.7)[:300]
    assert gpu_nms_batched(np.zeros((0, 4), np.float32), np.zeros((0, 3), np.float32), 0.3) == [[], [], []]


def test_nms_large_n_host_scan_path():
    dets, _ = _sorted(GI.nms_case(33000, 42))                 # > 32768: bitmask to host + host scan (reference layout)
    keep = np.zeros(33000, np.int32)
    num = ctypes.c_int(0)
    _lib.call("mnc_nms_topk", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 33000, 5, 0.5, 500, 0)
    want = native.nms_sorted(dets, 0.5)[:500]
    assert np.array_equal(keep[:num.value], want)


def test_mv_direct_vs_reference_fixture(golden):
    from nms.mv import mv
    mc = GI.mv_case(8)
    rm, rb = mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    assert np.array_equal(rb, golden["mv_box"])
    assert np.array_equal(rm, golden["mv_mask"])
    em, eb = mv(mc["boxes"], mc["masks"], np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32),
                mc["H"], mc["W"])
    assert em.shape == (0, 1, 21, 21) and eb.shape == (0, 4)
    with pytest.raises(_lib.MncError):
        mv(mc["boxes"], mc["masks"], np.array([999], np.int32), np.array([1], np.int32), np.array([1.0], np.float32),
           mc["H"], mc["W"])
    # the void symbol with the reference's exact 15-argument signature (lib/nms/gpu_mv.hpp:1-4), called as gpu_mv.pyx:27-30 does
    lib = _lib.load()
    boxes, masks = np.ascontiguousarray(mc["boxes"], np.float32), np.ascontiguousarray(mc["masks"], np.float32)
    inds, start = np.ascontiguousarray(mc["inds"], np.int32), np.ascontiguousarray(mc["start"], np.int32)
    wts = np.ascontiguousarray(mc["weights"], np.float32)
    R = len(start)
    om, ob = np.full((R, 1, 21, 21), -1, np.float32), np.full((R, 4), -1, np.int32)
    lib._mv(_lib.ptr(boxes), _lib.ptr(masks), boxes.shape[0], _lib.ptr(inds), _lib.ptr(start), _lib.ptr(wts), len(inds),
            mc["H"], mc["W"], boxes.shape[1], 21, R, _lib.ptr(om), _lib.ptr(ob), 0)
    assert np.array_equal(ob, golden["mv_box"]) and np.array_equal(om, golden["mv_mask"])


@pytest.mark.parametrize("tag", ["small", "full"])
def test_gpu_mask_voting_vs_reference_fixture(golden, tag):
    """Product host code (transform.mask_transform) + HIP nms + HIP mv, against the reference's python + .cu output."""
    from transform.mask_transform import gpu_mask_voting
    n, H, W, seed = GI.VOTING_CASES[tag]
    vc = GI.voting_case(n, H, W, seed)
    lm, lb = gpu_mask_voting(vc["masks"], vc["boxes"], vc["scores"], 21, 100, W, H)
    assert len(lm) == len(lb) == 20
    assert np.array_equal(np.array([len(b) for b in lb]), golden["vote_%s_count" % tag])
    assert np.array_equal(np.concatenate(lb, 0), golden["vote_%s_box" % tag])
    assert np.array_equal(np.concatenate(lm, 0), golden["vote_%s_mask" % tag])


def test_fused_voting_equals_step_by_step_composition():
    """mnc_mask_voting (fused) == the reference's composition of nms x20 + bbox_overlaps + mv, on the product's own
    modular path and on the oracle, at BASELINE's 600 instances / 600x1000 canvas and on float64 boxes (generic path)."""
    from transform import mask_transform as mt
    vc = GI.voting_case(600, 600, 1000, 23)
    fused = mt.gpu_mask_voting(vc["masks"], vc["boxes"], vc["scores"], 21, 100, 1000, 600)
    generic = mt.gpu_mask_voting(vc["masks"], vc["boxes"].astype(np.float64), vc["scores"], 21, 100, 1000, 600)
    from oracle import host as ohost
    want = ohost.gpu_mask_voting(vc["masks"], vc["boxes"], vc["scores"], 21, 100, 1000, 600)
    for got in (fused, generic):
        assert [len(b) for b in got[1]] == [len(b) for b in want[1]]
        assert np.array_equal(np.concatenate(got[1], 0), np.concatenate(want[1], 0))
        assert np.array_equal(np.concatenate(got[0], 0), np.concatenate(want[0], 0))
    assert fused[1][0].dtype == np.float64
    small = mt.gpu_mask_voting(vc["masks"][:3], vc["boxes"][:3], vc["scores"][:3], 21, 100, 1000, 600)
    osmall = ohost.gpu_mask_voting(vc["masks"][:3], vc["boxes"][:3], vc["scores"][:3], 21, 100, 1000, 600)
    assert np.array_equal(np.concatenate(small[1], 0), np.concatenate(osmall[1], 0))


@pytest.mark.parametrize("n,seed", [(600, 1), (67, 2), (1500, 3), (4500, 4)])
def test_fused_voting_device_order_ties_and_sizes(n, seed):
    """The library orders each class on the device (LDS bitonic sort, n <= 4096) or on the host (larger n) exactly as
    np.argsort(-s, kind="stable"): scores quantised to 1/16 (hundreds of exact ties per class), signed zeros, and sizes
    on both sides of the sort's power-of-two padding and of the device/host switch.  Also: an explicit `order` argument
    (the caller's own argsort) gives the same result as the library's."""
    import ctypes
    from mnc_amd import _lib
    from oracle import host as ohost
    from transform import mask_transform as mt
    vc = GI.voting_case(n, 300, 400, seed)
    scores = (np.round(vc["scores"] * 16) / 16).astype(np.float32)
    scores[::7, 3] = -0.0
    scores[1::7, 3] = 0.0
    got = mt.gpu_mask_voting(vc["masks"], vc["boxes"], scores, 21, 100, 400, 300)
    want = ohost.gpu_mask_voting(vc["masks"], vc["boxes"], scores, 21, 100, 400, 300)
    assert [len(b) for b in got[1]] == [len(b) for b in want[1]]
    assert np.array_equal(np.concatenate(got[1], 0), np.concatenate(want[1], 0))
    # (rows whose member scores all quantise to 0 have weights 0/0 = NaN in the reference too: equal_nan)
    assert np.array_equal(np.concatenate(got[0], 0), np.concatenate(want[0], 0), equal_nan=True)
    order = np.stack([np.argsort(-scores[:, c + 1], kind="stable") for c in range(20)]).astype(np.int32)
    cap = 20 * min(100, n)
    om, ob, osc = np.zeros((cap, 1, 21, 21), np.float32), np.zeros((cap, 4), np.int32), np.zeros(cap, np.float32)
    cnt, R = np.zeros(20, np.int32), ctypes.c_int(0)
    boxes, masks = np.ascontiguousarray(vc["boxes"], np.float32), np.ascontiguousarray(vc["masks"], np.float32)
    _lib.call("mnc_mask_voting", _lib.ptr(boxes), _lib.ptr(masks), _lib.ptr(scores), _lib.ptr(order), n, 21, 21, 100,
              float(ohost.MASK_MERGE_NMS_THRESH), float(ohost.MASK_MERGE_IOU_THRESH), 300, 400, _lib.ptr(om), _lib.ptr(ob),
              _lib.ptr(osc), _lib.ptr(cnt), ctypes.addressof(R), 0)
    assert list(cnt) == [len(b) for b in want[1]]
    assert np.array_equal(om[:R.value], np.concatenate(want[0], 0), equal_nan=True)
    assert np.array_equal(ob[:R.value], np.concatenate(want[1], 0)[:, :4].astype(np.int32))


def test_mv_many_candidates_and_properties():
    """Size-independent properties at BASELINE's 600 instances / 600x1000 canvas: (a) a result whose candidate list
    has > 1024 entries takes the global-memory path and still matches the oracle; (b) duplicating a result duplicates
    its output; (c) scaling is NOT linear (binarisation) but weights that sum to 1 over identical masks reproduce the
    single mask."""
    from nms.mv import mv
    vc = GI.voting_case(600, 600, 1000, 77)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 600, 1500).astype(np.int32)
    w = rng.uniform(0.1, 1, 1500).astype(np.float32)
    w /= w.sum()
    same = np.array([7, 7, 7], np.int32)
    inds = np.concatenate([big, same, big]).astype(np.int32)
    wts = np.concatenate([w, np.array([0.25, 0.25, 0.5], np.float32), w]).astype(np.float32)
    start = np.array([1500, 1503, 3003], np.int32)
    rm, rb = mv(vc["boxes"], vc["masks"], inds, start, wts, 600, 1000)
    om, ob = native.mv(vc["boxes"], vc["masks"], inds, start, wts, 600, 1000)
    assert np.array_equal(rb, ob) and np.array_equal(rm, om)
    assert np.array_equal(rm[0], rm[2]) and np.array_equal(rb[0], rb[2])
    one_m, one_b = mv(vc["boxes"], vc["masks"], np.array([7], np.int32), np.array([1], np.int32),
                      np.array([1.0], np.float32), 600, 1000)
    assert np.array_equal(one_b[0], rb[1])


@pytest.mark.skipif(not native.ref_available(), reason="oracle/_ref/libmnc_ref.so not present")
def test_against_reference_cu_code_directly():
    from nms.mv import mv
    from nms.gpu_nms import gpu_nms
    d = GI.nms_case(3000, 91)
    assert gpu_nms(d, 0.6, 0) == native.ref_gpu_nms(d, 0.6)
    mc = GI.mv_case(92, H=200, W=320)
    a = mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    b = native.ref_mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_bbox_overlaps_c_abi():
    from utils.cython_bbox import bbox_overlaps
    rng = np.random.default_rng(3)
    a = GI._boxes(rng, 600, 1000, 600).astype(np.float64)
    q = GI._boxes(rng, 5, 1000, 600).astype(np.float64)
    assert np.array_equal(bbox_overlaps(a, q), native.bbox_overlaps(a, q))


@pytest.mark.parametrize("n,thr", [(6000, 0.7), (600, 0.3)])
def test_nms_properties_at_full_size(n, thr):
    """Size-independent properties at the RPN's 6000 candidates / the voting's 600 instances: kept indices are in descending
    score order, no two kept boxes overlap by more than the threshold, every suppressed box overlaps an earlier kept one, and
    NMS of the kept set keeps everything (idempotence)."""
    from nms.gpu_nms import gpu_nms
    from oracle import native as onative
    dets = GI.nms_case(n, 77)
    keep = np.array(gpu_nms(dets, thr))
    assert len(keep) and np.all(np.diff(dets[keep, 4]) <= 0)
    iou = onative.bbox_overlaps(dets[keep, :4].astype(np.float64), dets[keep, :4].astype(np.float64))
    np.fill_diagonal(iou, 0)
    assert iou.max() <= thr + 1e-6
    order = np.argsort(-dets[:, 4], kind="stable")
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    dropped = np.setdiff1d(np.arange(n), keep)
    ov = onative.bbox_overlaps(dets[dropped, :4].astype(np.float64), dets[keep, :4].astype(np.float64))
    earlier = rank[keep][None, :] < rank[dropped][:, None]
    assert np.all(((ov > thr - 1e-6) & earlier).any(1))
    again = gpu_nms(dets[keep], thr)
    assert list(again) == list(range(len(keep)))
