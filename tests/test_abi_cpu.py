"""CPU: libmnc_hip.so loads without a GPU, exports every symbol include/mnc_hip.h declares, marshals arguments, and
fails loudly (status + message) instead of computing anything when there is no device.  No compute calls here except
mnc_bbox_overlaps, which is a host function in the reference too (lib/utils/bbox.pyx)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import mnc_amd
from mnc_amd import _build, _lib

mnc_amd.install_paths()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    decls = _lib.parse_header()
    assert len(decls) >= 40
    for name in decls:
        assert hasattr(lib, name), name
    for must in ("_nms", "_mv", "mnc_nms", "mnc_mv", "mnc_bbox_overlaps", "mnc_conv3x3", "mnc_fc", "mnc_roi_warp"):
        assert must in decls
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(decls) <= exported
    assert lib.mnc_version().startswith(b"mnc_hip")


REF_NMS_DIR = "/root/reference/lib/nms"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_ref_binding_program(exe):
    """tests/c/ref_binding_main.cpp -> exe: a C++ TU that knows `_nms` / `_mv` only from the reference's headers (the real ones when
    /root/reference is mounted, their two declarations verbatim otherwise), linked against libmnc_hip.so."""
    cmd = ["g++", "-O1", "-std=c++11", os.path.join(REPO, "tests", "c", "ref_binding_main.cpp"), "-o", exe,
           "-L", os.path.join(REPO, "mnc_amd"), "-lmnc_hip", "-Wl,-rpath," + os.path.join(REPO, "mnc_amd"),
           "-Wl,-rpath-link,/opt/rocm/lib"]
    if os.path.isfile(os.path.join(REF_NMS_DIR, "gpu_nms.hpp")) and os.path.isfile(os.path.join(REF_NMS_DIR, "gpu_mv.hpp")):
        cmd += ["-DMNC_REF_HEADERS", "-I", REF_NMS_DIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the reference's C++ binding does not link against libmnc_hip.so:\n" + r.stderr
    return exe


def test_reference_cxx_binding_links(tmp_path):
    """b1/b2 at link level: the reference's extensions are C++ (lib/setup.py:126-130, 143-147 language='c++';
    gpu_nms.pyx:13-14, gpu_mv.pyx:7-8 `cdef extern from "gpu_nms.hpp"`), so they import the MANGLED names -- the ones
    oracle/_ref/libmnc_ref.so (the reference's own .cu files) exports.  libmnc_hip.so must export the same two."""
    _lib.load()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    want = {"_Z4_nmsPiS_PKfiifi", "_Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii"}
    assert want <= exported
    ref_so = os.path.join(REPO, "oracle", "_ref", "libmnc_ref.so")
    if os.path.isfile(ref_so):          # what the reference's own sources export under the same compiler ABI
        ref = subprocess.run(["nm", "-D", "--defined-only", ref_so], capture_output=True, text=True).stdout
        assert want <= {l.split()[-1] for l in ref.splitlines() if " T " in l}
    exe = build_ref_binding_program(str(tmp_path / "ref_binding_main"))
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "_Z4_nmsPiS_PKfiifi" in und and "_Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii" in und      # bound to the mangled exports
    r = subprocess.run([exe, "link"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "_nms 0x" in r.stdout, r.stdout + r.stderr


def test_reference_signatures_are_kept():
    """b1/b2: `_nms` (gpu_nms.hpp:1-2) has 7 parameters, `_mv` (gpu_mv.hpp:1-4) has 15, in the reference's order."""
    d = _lib.parse_header()
    assert d["_nms"][2] == ["keep_out", "num_out", "boxes_host", "boxes_num", "boxes_dim", "nms_overlap_thresh", "device_id"]
    assert d["_mv"][2] == ["all_boxes", "all_masks", "all_boxes_num", "candidate_inds", "candidate_start",
                           "candidate_weights", "candidate_num", "image_height", "image_width", "box_dim", "mask_size",
                           "result_num", "finalize_output_mask", "finalize_output_box", "device_id"]
    assert d["_nms"][0] is None and d["_mv"][0] is None


@pytest.mark.skipif(_lib.device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_errors_not_fallbacks():
    h = ctypes.c_void_p()
    with pytest.raises(_lib.MncError) as e:
        _lib.call("mnc_ctx_create", ctypes.addressof(h), 0)
    assert e.value.code in (1, 2) and str(e.value)
    keep = np.zeros(4, np.int32)
    num = ctypes.c_int(7)
    dets = np.array([[0, 0, 10, 10, 0.9], [1, 1, 11, 11, 0.8]], np.float32)
    with pytest.raises(_lib.MncError):
        _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), _lib.ptr(dets), 2, 5, 0.5, 0)
    from nms.gpu_nms import gpu_nms
    with pytest.raises(_lib.MncError):
        gpu_nms(dets, 0.5)
    from mnc_amd.engine import Net
    from mnc_amd import models
    with pytest.raises(RuntimeError):
        Net(models.write_mnc_5stage_test_prototxt(width_div=8), {}, 1)
    # n == 0 short-circuits before any device work, as nms_wrapper.py:16-17 does
    _lib.call("mnc_nms", _lib.ptr(keep), ctypes.addressof(num), None, 0, 5, 0.5, 0)
    assert num.value == 0
    from nms.nms_wrapper import nms
    assert nms(np.zeros((0, 5), np.float32), 0.3) == []


def test_bbox_overlaps_host_function():
    from utils.cython_bbox import bbox_overlaps
    from oracle import native
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 500, (600, 4))
    a[:, 2:] += a[:, :2]
    q = a[:5] + 3.0
    got = bbox_overlaps(a, q)
    assert got.shape == (600, 5) and np.array_equal(got, native.bbox_overlaps(a, q))
    assert bbox_overlaps(np.zeros((0, 4)), q).shape == (0, 5)
    with pytest.raises(ValueError):
        bbox_overlaps(np.zeros((3, 2)), q)


def test_build_is_up_to_date_and_product_never_imports_the_oracle():
    assert _build.up_to_date()
    root = os.path.join(os.path.dirname(_lib.HERE), "mnc_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libmnc_oracle" not in text, f
    for f in ("tools/demo.py", "tools/_init_paths.py"):
        text = open(os.path.join(os.path.dirname(_lib.HERE), f)).read()
        assert "oracle" not in text


def test_bench_roofline_by_kernel_groups_shapes_and_tags_traffic(tmp_path, monkeypatch):
    """bench.roofline_by_kernel: the InnerProduct scope is split by shape (algorithmic flop), the Winograd scope carries executed
    flop next to algorithmic, conv1_1 is priced against HBM; counter traffic is reported only from a PMC profile whose build tag
    equals the running build's source hash, per position of the InnerProduct kernel's launch cycle."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    from mnc_amd import _build
    rec = []
    for _ in range(3):                                      # three event steps
        rec += [("fc_mfma", 0.14, 15.4140672e9, 223.5e6), ("fc_mfma", 0.47, 61.6562688e9, 446.1e6), ("fc_mfma", 0.09, 10.0663296e9, 76.9e6),
                ("fc_mfma", 0.47, 61.6562688e9, 446.1e6), ("fc_mfma", 0.09, 10.0663296e9, 76.9e6),
                ("conv3x3_wino_mfma", 0.2, 44.2368e9, 307.3e6), ("conv3x3_c3", 0.064, 2.0736e9, 160.8e6)]
    prof = {"_build": "stale", "fc_mfma_dma16_kernel<10>": {"calls": 15, "hbm_bytes_corrected": 3.0e8,
            "by_position": [{"hbm_bytes_corrected": 2.6e8}, {"hbm_bytes_corrected": 4.8e8}, {"hbm_bytes_corrected": 1.2e8},
                            {"hbm_bytes_corrected": 4.8e8}, {"hbm_bytes_corrected": 1.2e8}] * 2},
            "conv3x3_c3_kernel<0>": {"calls": 3, "hbm_bytes_corrected": 1.7e8}}
    os.makedirs(str(tmp_path / "profiles"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    with open(str(tmp_path / "profiles" / "pmc_latest.json"), "w") as f:
        json.dump(prof, f)
    rows = bench.roofline_by_kernel(rec, 3)
    what = {r["what"]: r for r in rows}
    assert set(what) == {"fc6_maskest (300 x 256 x 100352)", "fc6 / fc6_mask (300 x 4096 x 25088)", "fc7 / fc7_mask (300 x 4096 x 4096)",
                         "conv3x3_wino_mfma", "conv3x3_wino_mfma (all launches of an image)", "conv3x3_c3"}
    fc6 = what["fc6 / fc6_mask (300 x 4096 x 25088)"]
    assert fc6["launches_per_image"] == 2.0 and abs(fc6["frac"] - 61.6562688e9 / 0.47e-3 / 1e12 / 157.3) < 1e-9
    assert all(r.get("traffic") is None and r.get("traffic_mb_per_image") is None for r in rows)   # wrong build: no traffic
    assert "stale" in fc6["traffic_source"]
    w = what["conv3x3_wino_mfma"]
    assert abs(w["executed_gflop_per_launch"] - 44.2368 / 2.25) < 1e-9 and w["executed_frac_of_peak"] < w["frac"]
    assert what["conv3x3_c3"]["bound"] == "hbm" and what["conv3x3_c3"]["unit"] == "GB/s"
    prof["_build"] = _build.source_hash()
    with open(str(tmp_path / "profiles" / "pmc_latest.json"), "w") as f:
        json.dump(prof, f)
    what = {r["what"]: r for r in bench.roofline_by_kernel(rec, 3)}
    assert what["fc6 / fc6_mask (300 x 4096 x 25088)"]["traffic"] == 4.8e8                       # positions 1, 3, 6, 8
    assert what["fc7 / fc7_mask (300 x 4096 x 4096)"]["traffic"] == 1.2e8
    assert what["fc6_maskest (300 x 256 x 100352)"]["traffic"] == 2.6e8
    assert abs(what["fc7 / fc7_mask (300 x 4096 x 4096)"]["traffic_over_algorithmic"] - 1.2e8 / 76.9e6) < 1e-9
    assert what["conv3x3_c3"]["traffic"] == 1.7e8


def test_tuning_values_are_per_context_and_checked():
    """mnc_ctx_set_tuning is declared and exported; keys are validated by name (no context needed to check the table)."""
    d = _lib.parse_header()
    assert d["mnc_ctx_set_tuning"][2] == ["ctx", "name", "value"]
    src = open(os.path.join(REPO, "mnc_amd", "csrc", "mnc_internal.h")).read()
    for key in ("FC_TILE", "FC_HALF", "FCX3_WIDE", "WINO_ROWS", "ROI_WARP_VARIANT", "TOPK_SINGLE_WG"):
        assert "X(%s)" % key in src
    # no launch path reads the environment: getenv appears in ctx.hip (context creation) only
    import glob
    users = [os.path.basename(p) for p in glob.glob(os.path.join(REPO, "mnc_amd", "csrc", "*.hip")) if "getenv(" in open(p).read()]
    assert users == ["ctx.hip"], users


def test_no_product_kernel_spills_registers():
    """every kernel of the product library's gfx950 code objects has .vgpr_spill_count == 0 and no scratch
    (private segment) at all -- read from the code-object notes (llvm-readelf), no GPU needed.  Round 5 shipped four offenders
    (roi_warp_row_kernel<1, 1> / <1, 2>, conv2d_c8_kernel<2>, conv3x3_wino2_kernel<1, 0, 0, 1>)."""
    import importlib.util
    if not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no llvm-readelf in this image")
    _lib.load()
    if b"tuning" in _lib.load().mnc_version():
        pytest.skip("tuning build: measurement kernels are exempt")
    spec = importlib.util.spec_from_file_location("spill_report", os.path.join(REPO, "tools", "spill_report.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    notes = mod.kernel_notes(_lib.LIB_PATH)
    assert len(notes) > 150, len(notes)
    assert any("conv3x3_sw_kernel" in k for k in notes) and any("fc_mfma_dma16_kernel" in k for k in notes)
    bad = {k: v for k, v in notes.items() if v["vgpr_spill"] or v["scratch"]}
    assert not bad, bad
