"""TEST INFRASTRUCTURE ONLY -- ctypes access to the CPU oracle libraries.

  libmnc_oracle.so      : oracle/mnc_oracle.c, our C restatement (kind "port")
  _ref/libmnc_ref.so    : the reference's own lib/nms/{nms,mv}_kernel.cu compiled for the CPU by
                          oracle/build_ref.py (kind "reference"); prebuilt, it also travels to the GPU box.

Nothing under mnc_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(HERE, "libmnc_oracle.so")
_REF_SO = os.path.join(HERE, "_ref", "libmnc_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)
_u64p = ctypes.POINTER(ctypes.c_ulonglong)


def build(force=False):
    """Compile the C restatement (always) and the reference build (only where /root/reference exists)."""
    src = os.path.join(HERE, "mnc_oracle.c")
    if force or not os.path.isfile(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp",
                        "-fvisibility=hidden", "-o", _ORACLE_SO, src, "-lm"], check=True)
    from . import build_ref
    build_ref.build(force=False)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_ORACLE_SO):
            build()
        _lib = ctypes.CDLL(_ORACLE_SO)
    return _lib


def ref_available():
    return os.path.isfile(_REF_SO)


def ref():
    """The reference's own _nms/_mv (C++-mangled symbols from the .cu files)."""
    global _ref
    if _ref is None:
        if not os.path.isfile(_REF_SO):
            from . import build_ref
            if build_ref.build() is None:
                raise RuntimeError("oracle/_ref/libmnc_ref.so missing and /root/reference not mounted")
        _ref = ctypes.CDLL(_REF_SO)
    return _ref


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---- our restatement ----------------------------------------------------------------------------
def nms_sorted(boxes_sorted, thresh):
    """orc_nms: boxes already sorted by descending score; returns indices into the sorted array."""
    b = _c(boxes_sorted, np.float32)
    n, dim = b.shape
    keep = np.zeros(max(n, 1), np.int32)
    num = ctypes.c_int(0)
    lib().orc_nms(_p(keep, _i32p), ctypes.byref(num), _p(b, _f32p), n, dim, ctypes.c_float(thresh))
    return keep[:num.value].copy()


def nms_mask(boxes_sorted, thresh):
    b = _c(boxes_sorted, np.float32)
    n, dim = b.shape
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), np.uint64)
    lib().orc_nms_mask(_p(b, _f32p), n, dim, ctypes.c_float(thresh), _p(mask, _u64p))
    return mask


def gpu_nms(dets, thresh):
    """Restates lib/nms/gpu_nms.pyx:16-31 on top of orc_nms.  Tie order pinned (score desc, index asc) where the
    reference's `argsort()[::-1]` leaves it unspecified; identical for distinct scores."""
    dets = _c(dets, np.float32)
    order = np.argsort(-dets[:, 4], kind="stable")
    keep = nms_sorted(dets[order, :], thresh)
    return [int(i) for i in order[keep]]


def bbox_overlaps(boxes, query):
    b, q = _c(boxes, np.float64), _c(query, np.float64)
    out = np.zeros((b.shape[0], q.shape[0]), np.float64)
    lib().orc_bbox_overlaps(_p(b, _f64p), b.shape[0], _p(q, _f64p), q.shape[0], _p(out, _f64p))
    return out


def _mv_call(fn, all_boxes, all_masks, cand_inds, cand_start, cand_weights, H, W, with_device_id):
    boxes = _c(all_boxes, np.float32)
    masks = _c(all_masks, np.float32)
    inds, start, wts = _c(cand_inds, np.int32), _c(cand_start, np.int32), _c(cand_weights, np.float32)
    S = masks.shape[3]
    R = start.shape[0]
    out_mask = np.zeros((R, 1, S, S), np.float32)
    out_box = np.zeros((R, boxes.shape[1]), np.int32)
    if R == 0:
        return out_mask, out_box
    args = [_p(boxes, _f32p), _p(masks, _f32p), boxes.shape[0], _p(inds, _i32p), _p(start, _i32p),
            _p(wts, _f32p), inds.shape[0], int(H), int(W), boxes.shape[1], S, R,
            _p(out_mask, _f32p), _p(out_box, _i32p)]
    if with_device_id:
        args.append(0)
    fn(*args)
    return out_mask, out_box


def mv(all_boxes, all_masks, cand_inds, cand_start, cand_weights, H, W):
    """Restates lib/nms/gpu_mv.pyx:13-31 on top of orc_mv."""
    return _mv_call(lib().orc_mv, all_boxes, all_masks, cand_inds, cand_start, cand_weights, H, W, False)


# The SPEC-CHOICEs of the three unpinned layers as switches (oracle/SPEC.md section 6; the same names and values as
# `mnc_layer_conventions` in include/mnc_hip.h / mnc_amd.engine.LAYER_CONVENTIONS).  All defaults = the SPEC.
SPEC_CONVENTIONS = {"warp_sample": 0, "warp_round_edges": 0, "warp_no_plus_one": 0, "warp_oob": 0, "resize_mode": 0,
                    "maskpool_binary": 0, "maskpool_thresh": 0.4}
_conv = dict(SPEC_CONVENTIONS)


class conventions(object):
    """with native.conventions(warp_sample=1): ...  -- roi_warp / mask_resize / mask_pool (and with them oracle.net.head)
    evaluate the alternative convention inside the block."""

    def __init__(self, **kw):
        bad = set(kw) - set(SPEC_CONVENTIONS)
        if bad:
            raise KeyError("unknown layer convention(s): %s" % sorted(bad))
        self.kw = kw

    def __enter__(self):
        self.saved = dict(_conv)
        _conv.update(self.kw)
        return self

    def __exit__(self, *exc):
        _conv.clear()
        _conv.update(self.saved)


def roi_warp(feat, rois, PH, PW, scale):
    f = _c(feat, np.float32)
    if f.ndim == 4:
        f = f[0]
    r = _c(rois, np.float32)
    C, H, W = f.shape
    out = np.zeros((r.shape[0], C, PH, PW), np.float32)
    lib().orc_roi_warp_ex(_p(f, _f32p), C, H, W, _p(r, _f32p), r.shape[0], PH, PW, ctypes.c_float(scale),
                          int(_conv["warp_sample"]), int(_conv["warp_round_edges"]), int(_conv["warp_no_plus_one"]),
                          int(_conv["warp_oob"]), _p(out, _f32p))
    return out


def roi_pool(feat, rois, PH, PW, scale):
    """Caffe ROIPooling: feat [N,C,H,W], rois [R,5] (batch index first) -> [R,C,PH,PW]."""
    f, r = _c(feat, np.float32), _c(rois, np.float32)
    N, C, H, W = f.shape
    out = np.zeros((r.shape[0], C, PH, PW), np.float32)
    lib().orc_roi_pool.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_float, _f32p]
    lib().orc_roi_pool(_p(f, _f32p), N, C, H, W, _p(r, _f32p), r.shape[0], PH, PW, scale, _p(out, _f32p))
    return out


def mask_resize(mask, OH, OW):
    m = _c(mask, np.float32)
    R, _, IH, IW = m.shape
    out = np.zeros((R, 1, OH, OW), np.float32)
    lib().orc_mask_resize_ex(_p(m, _f32p), R, IH, IW, OH, OW, int(_conv["resize_mode"]), _p(out, _f32p))
    return out


def mask_pool(feat, mask):
    f, m = _c(feat, np.float32), _c(mask, np.float32)
    R, C, H, W = f.shape
    out = np.zeros_like(f)
    lib().orc_mask_pool_ex(_p(f, _f32p), _p(m, _f32p), R, C, H, W, int(_conv["maskpool_binary"]),
                           ctypes.c_float(_conv["maskpool_thresh"]), _p(out, _f32p))
    return out


def maxpool2(x):
    a = _c(x, np.float32)
    H, W = a.shape[-2:]
    planes = int(np.prod(a.shape[:-2]))
    OH, OW = (H - 2 + 1) // 2 + 1, (W - 2 + 1) // 2 + 1
    out = np.zeros(a.shape[:-2] + (OH, OW), np.float32)
    lib().orc_maxpool2.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, _f32p]
    lib().orc_maxpool2(_p(a, _f32p), planes, H, W, _p(out, _f32p))
    return out


# ---- the reference's own code (oracle/_ref) -----------------------------------------------------
def ref_nms_sorted(boxes_sorted, thresh):
    """lib/nms/nms_kernel.cu `_nms` itself (gpu_nms.hpp:1-2), run on the CPU."""
    b = _c(boxes_sorted, np.float32)
    n, dim = b.shape
    keep = np.zeros(max(n, 1), np.int32)
    num = ctypes.c_int(0)
    fn = getattr(ref(), "_Z4_nmsPiS_PKfiifi")
    fn(_p(keep, _i32p), ctypes.byref(num), _p(b, _f32p), n, dim, ctypes.c_float(thresh), 0)
    return keep[:num.value].copy()


def ref_gpu_nms(dets, thresh):
    dets = _c(dets, np.float32)
    order = dets[:, 4].argsort()[::-1]
    keep = ref_nms_sorted(dets[order, :], thresh)
    return [int(i) for i in order[keep]]


def ref_mv(all_boxes, all_masks, cand_inds, cand_start, cand_weights, H, W):
    """lib/nms/mv_kernel.cu `_mv` itself (gpu_mv.hpp:1-4), run on the CPU.  Allocates N*H*W floats."""
    fn = getattr(ref(), "_Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii")
    return _mv_call(fn, all_boxes, all_masks, cand_inds, cand_start, cand_weights, H, W, True)
