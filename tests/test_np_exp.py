"""mnc_amd/csrc/np_exp.h restates numpy's float32 exp loop (the arithmetic behind `np.exp(dw) * widths` in the reference's
bbox_transform_inv, lib/transform/bbox_transform.py:88-89).  The header is compiled for the host and compared bit for bit with
np.exp over the float32 encoding space, so the device-resident ProposalLayer / StageBridgeLayer can be held to array_equal
against the numpy Python layers (tests/test_gpu_engine.py)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("npexp") / "np_exp_shim.so")
    subprocess.check_call([cxx, "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(HERE, "np_exp_shim.cpp"),
                           "-o", so])
    lib = ctypes.CDLL(so)
    lib.np_exp_f32_array.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    lib.decode_box.argtypes = [ctypes.c_void_p] * 3
    return lib


def _numpy_uses_its_simd_exp():
    """numpy's own float32 exp differs from the correctly rounded value on ~39 % of inputs; libm's expf on < 1 %."""
    x = np.linspace(-3, 3, 4001, dtype=np.float32)
    return float((np.exp(x) != np.exp(x.astype(np.float64)).astype(np.float32)).mean()) > 0.1


def _mine(lib, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib.np_exp_f32_array(x.ctypes.data, out.ctypes.data, x.size)
    return out


def test_np_exp_restatement_is_bit_identical_to_numpy(shim):
    if not _numpy_uses_its_simd_exp():
        pytest.skip("this numpy build computes float32 exp with libm (no AVX2/AVX512F dispatch)")
    bad = total = 0
    with np.errstate(all="ignore"):
        for hi in range(0, 65536, 64):                    # 1024 chunks x 65536 consecutive bit patterns, all exponents/signs
            x = (np.arange(65536, dtype=np.uint32) + np.uint32(hi << 16)).view(np.float32)
            want, got = np.exp(x), _mine(shim, x)
            ne = (want.view(np.uint32) != got.view(np.uint32)) & ~(np.isnan(want) & np.isnan(got))
            bad += int(ne.sum())
            total += x.size
    assert total == 1 << 26 and bad == 0
    # the range the path feeds it (box regression deltas), dense, through the strided views the reference takes (deltas[:, 2::4])
    rng = np.random.default_rng(0)
    d = (rng.standard_normal((500000, 4)) * 0.7).astype(np.float32)
    assert np.array_equal(np.exp(d[:, 2::4]).ravel(), _mine(shim, d[:, 2]))
    assert np.array_equal(np.exp(d[:, 3::4]).ravel(), _mine(shim, d[:, 3]))


def test_decode_in_kernel_order_equals_bbox_transform_inv(shim):
    """The kernels' float32 expression order around the exp (widths, centres, +-0.5 * w) reproduces the product's / the
    reference's numpy bbox_transform_inv exactly."""
    if not _numpy_uses_its_simd_exp():
        pytest.skip("this numpy build computes float32 exp with libm")
    import mnc_amd
    mnc_amd.install_paths()
    from transform.bbox_transform import bbox_transform_inv
    rng = np.random.default_rng(3)
    n = 20000
    xy = rng.uniform(-200, 1100, (n, 2)).astype(np.float32)
    wh = rng.uniform(1, 700, (n, 2)).astype(np.float32)
    boxes = np.hstack((xy, xy + wh)).astype(np.float32)
    deltas = (rng.standard_normal((n, 4)) * np.array([0.3, 0.3, 0.6, 0.6])).astype(np.float32)
    want = bbox_transform_inv(boxes, deltas)
    got = np.empty((n, 4), np.float32)
    for i in range(n):
        shim.decode_box(boxes[i].ctypes.data, deltas[i].ctypes.data, got[i].ctypes.data)
    assert want.dtype == np.float32 and np.array_equal(want, got)
