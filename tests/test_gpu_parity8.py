"""GPU: the BENCHMARKED executor -- the one-call native pipeline (mnc_forward_image, csrc/pipeline.hip) -- against the CPU oracle
DIRECTLY (not through the Python engine), at BASELINE's full size (600x1000, VGG-16 widths, 300 RoIs per stage), on all eight
BASELINE images (seeds 0..7), in fp32, in bf16x3 and in f16 (the last with its own 5e-3 bar).

Two protocols per image:

  teacher-forced  every data-dependent hop is checked on the device's own inputs (tests/test_gpu_engine.py:check_forward):
                  prepared input == oracle prep (bit-exact); conv5_3 / RPN blobs vs oracle trunk (tolerance);
                  rois == oracle ProposalLayer on the device's RPN blobs (bit-exact); stage 2/3 head outputs vs oracle head on the
                  device's rois (tolerance); rois_ext == oracle StageBridge on the device's scores (bit-exact); stage 4/5 likewise;
                  boxes == oracle im_detect tail (bit-exact); voted instances == oracle gpu_mask_voting on the device's
                  boxes / masks / scores (bit-exact).
  free-running    the oracle runs the whole image on its own (its own rois, its own arg-max classes); the two end results are
                  compared as a user would: `rois` rows the two lists share, final instances matched by class and box, score / box /
                  mask differences on the matched ones.  The path is discontinuous in places (IoU > 0.7 decisions, arg-max class,
                  mv's rounding of candidate boxes to canvas pixels and its `> 0.4` bounds): a 1e-6 upstream difference can flip
                  one of them, after which a row or a few mask cells legitimately differ by O(0.1) -- the statistic is RECORDED
                  (gpurun_out/parity_report.txt -> profiles/r03_parity_report*.txt), with loose floors, not held to equality.

The oracle trunk of an image is computed once and shared by both math modes."""
import os

import numpy as np
import pytest

import mnc_amd
from gpu_util import err, from_c8
from mnc_amd import models, synth
from mnc_amd.native_net import NativeNet
from oracle import host as ohost
from oracle import net as onet
from test_gpu_engine import BF16_TOL, F16_TOL, FP32_TOL, MIXED_TOL, NUMPY_SIMD_EXP, X3_TOL, _log

pytestmark = pytest.mark.gpu
mnc_amd.install_paths()

SEEDS = tuple(range(8))
# Floors = the recorded figures (profiles/r04_parity_report*.txt; VERDICT r3: the old floors -- 295 rois, 97 % matched -- would also
# have passed a much worse run).  fp32 (F(4x4,3x3) trunk since round 4): 300 / 300 rois and 100 / 100 instances matched on every one
# of the eight images; mask cells off by > 1e-3: 0-2 of 44100 on every image with the builds since round 4's third (and round 5's:
# profiles/r05_parity_report*.txt).  Round 4's first two builds (d7ddce37de4ff8ec and its predecessor: the scalar-transform F(4x4)
# kernel, profiles/r04_parity_report_v1 / _v2.txt) recorded 392 on seed 0 -- one flipped `> 0.4` bound of mv moves a box edge by a
# pixel and the whole resampled mask of that ONE instance with it.  So the fp32 ceiling has two parts: at most 8 cells off per image
# outside the single worst instance (CEIL_CELLS_REST), and room for one flipped bound in that instance (CEIL_CELLS_OFF).
# bf16x3: 290-299 rois, 98-100 matched.  f16 does not claim the 1e-3 bar (no `rois` row within 0.01 px, 78-92 of 100 matched): its
# floors only guard against a collapse; the mode that does claim it is "mixed" (below).
FLOOR_ROIS = {"fp32": 300, "bf16x3": 288, "f16": 0, "mixed": 285, "bf16": 0}            # mean over the eight images
FLOOR_MATCHED = {"fp32": 1.0, "bf16x3": 0.97, "f16": 0.6, "mixed": 0.95, "bf16": 0.3}
CEIL_CELLS_OFF = {"fp32": 0.01, "bf16x3": 0.06, "f16": 1.0, "mixed": 0.15, "bf16": 1.0}  # fraction of the matched instances' mask cells, per image
CEIL_CELLS_REST = {"fp32": 8}    # cells off per image NOT counting the instance with the most (absent: not checked)
_cache = {}


@pytest.fixture(scope="module")
def vgg():
    path = models.write_mnc_5stage_test_prototxt()
    return synth.synthetic_weights(path, seed=0)


def _oracle_image(w, seed):
    """Everything of the oracle that does not depend on the device: input, trunk, RPN, its own rois and full result."""
    if seed not in _cache:
        im = np.random.default_rng(seed).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
        data, im_info, scale = ohost.prepare_mnc_args(im)
        ref = {}
        c5 = onet.trunk(w, data, ref)
        prob, bbox = onet.rpn(w, c5, ref)
        rois = ohost.proposal_forward(prob, bbox, im_info)
        h1 = onet.head(w, c5, rois, False)
        rois_ext = ohost.stage_bridge_forward_test(rois, h1["bbox_pred"], h1["seg_cls_prob"], im_info)
        h2 = onet.head(w, c5, rois_ext, True)
        boxes, masks, scores = ohost.im_detect_tail(rois, h1["mask_proposal"], h1["seg_cls_prob"], rois_ext, h2["mask_proposal"],
                                                    h2["seg_cls_prob"], scale, im.shape)
        lm, lb = ohost.gpu_mask_voting(masks, boxes, scores, 21, 100, im.shape[1], im.shape[0])
        _cache[seed] = dict(im=im, data=data, im_info=im_info, scale=scale, c5=c5, conv5_3=ref["conv5_3"], prob=prob, bbox=bbox,
                            rois=rois, rois_ext=rois_ext, lm=lm, lb=lb)
    return _cache[seed]


def _iou(a, b):
    x1, y1 = np.maximum(a[0], b[:, 0]), np.maximum(a[1], b[:, 1])
    x2, y2 = np.minimum(a[2], b[:, 2]), np.minimum(a[3], b[:, 3])
    inter = np.clip(x2 - x1 + 1, 0, None) * np.clip(y2 - y1 + 1, 0, None)
    return inter / ((a[2] - a[0] + 1) * (a[3] - a[1] + 1) + (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) - inter)


def _free_running(o, got_m, got_b, dev_rois, dev_rois_ext):
    """-> dict of the free-running agreement figures of one image."""
    # rois: device rows that are (within 0.01 px) a row of the oracle's list -- as a set: one flipped NMS decision shifts every
    # later row by one position without changing which boxes were kept.  Bit-equality is not expected in this protocol (the
    # device's rpn_bbox_pred differs from the oracle's by its 1e-6 tolerance; bit-equality on the SAME blobs is the teacher-forced
    # check above).  rois_ext: of those, rows whose stage-2 box is within 0.05 px of the oracle's box for the same proposal.
    d = np.abs(dev_rois[:, None, 1:] - o["rois"][None, :, 1:]).max(axis=2)
    j = d.argmin(axis=1)
    pairs = [(i, int(j[i])) for i in range(len(dev_rois)) if d[i, j[i]] < 0.01]
    same_rois = len(pairs)
    same_order = sum(1 for i, k in pairs if i == k)
    same_ext = sum(1 for i, k in pairs if float(np.abs(dev_rois_ext[i] - o["rois_ext"][k]).max()) < 0.05)
    n_dev = sum(len(b) for b in got_b)
    n_orc = sum(len(b) for b in o["lb"])
    matched, max_mask, max_score, max_box, cells, cells_off, sum_mask = 0, 0.0, 0.0, 0.0, 0, 0, 0.0
    worst_instance = 0
    for c in range(20):
        gb, ob = np.asarray(got_b[c], np.float64), np.asarray(o["lb"][c], np.float64)
        used = set()
        for i in range(len(gb)):
            if not len(ob):
                break
            iou = _iou(gb[i, :4], ob[:, :4])
            j = int(np.argmax(iou))
            if iou[j] >= 0.9 and j not in used:
                used.add(j)
                matched += 1
                dm = np.abs(np.asarray(got_m[c][i], np.float64) - o["lm"][c][j])
                dm = dm[np.isfinite(dm)]
                max_mask = max(max_mask, float(dm.max()) if dm.size else 0.0)
                cells += dm.size
                cells_off += int((dm > 1e-3).sum())
                worst_instance = max(worst_instance, int((dm > 1e-3).sum()))
                sum_mask += float(dm.sum())
                max_score = max(max_score, abs(gb[i, 4] - ob[j, 4]))
                max_box = max(max_box, float(np.abs(gb[i, :4] - ob[j, :4]).max()))
    return dict(same_rois=same_rois, same_order=same_order, same_rois_ext=same_ext, n_dev=n_dev, n_orc=n_orc, matched=matched,
                max_mask=max_mask, mean_mask=sum_mask / max(cells, 1), cells=cells, cells_off=cells_off,
                cells_off_rest=cells_off - worst_instance, max_score=max_score,
                max_box=max_box)


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "f16", "mixed", "bf16"])
def test_native_pipeline_vs_oracle_on_the_eight_baseline_images(vgg, math):
    w = vgg
    tol = {"fp32": FP32_TOL, "bf16x3": X3_TOL, "f16": F16_TOL, "mixed": MIXED_TOL, "bf16": BF16_TOL}[math]
    nat = NativeNet(w, math=math)
    K, R = 21, 300
    lines, stats = [], []
    try:
        for seed in (SEEDS[:2] if math == "bf16" else SEEDS):      # (plain bf16 claims nothing: two images record its figures)
            o = _oracle_image(w, seed)
            im, im_info = o["im"], o["im_info"]
            got_m, got_b = nat.detect(im)
            # ---- teacher-forced -------------------------------------------------------------------------------------------
            assert np.array_equal(nat.blob("data"), o["data"])                               # a1: device prep, bit-exact
            c5 = nat.blob("conv5_3")
            _, C, h, ww = c5.shape
            c5 = from_c8(c5.reshape(-1), C, h, ww)[None]
            prob, bbox = nat.blob("rpn_cls_prob_reshape"), nat.blob("rpn_bbox_pred")
            rep = [("conv5_3", err(c5, o["conv5_3"])), ("rpn_cls_prob_reshape", err(prob, o["prob"])),
                   ("rpn_bbox_pred", err(bbox, o["bbox"]))]
            rois, rois_ext = nat.blob("rois"), nat.blob("rois_ext")
            assert rois.shape == (R, 5)
            if NUMPY_SIMD_EXP:
                assert np.array_equal(rois, ohost.proposal_forward(prob, bbox, im_info)), seed   # a5-a8 on the device's RPN blobs
            hs = nat.blob("head_scores")
            masks, scores = nat.blob("mask_proposal"), nat.blob("seg_cls_prob")
            assert hs.shape == (2 * R, 6 * K) and masks.shape == (2 * R, 1, 21, 21) and scores.shape == (2 * R, K)
            h1 = onet.head(w, c5[0], rois, False)                                                # oracle head on the DEVICE's conv5_3 + rois
            rep += [("mask_proposal", err(masks[:R], h1["mask_proposal"])), ("seg_cls_prob", err(scores[:R], h1["seg_cls_prob"])),
                    ("cls_score", err(hs[:R, :K], h1["cls_score"])), ("bbox_pred", err(hs[:R, 2 * K:], h1["bbox_pred"]))]
            want_ext = ohost.stage_bridge_forward_test(rois, np.ascontiguousarray(hs[:R, 2 * K:]), scores[:R], im_info)
            if NUMPY_SIMD_EXP:
                assert np.array_equal(rois_ext, want_ext), seed                                  # a16 on the device's scores
            h2 = onet.head(w, c5[0], rois_ext, True)
            rep += [("mask_proposal_ext", err(masks[R:], h2["mask_proposal"])), ("seg_cls_prob_ext", err(scores[R:], h2["seg_cls_prob"])),
                    ("bbox_pred_ext", err(hs[R:, 2 * K:], h2["bbox_pred"]))]
            boxes = nat.blob("boxes")
            ob, _, _ = ohost.im_detect_tail(rois, masks[:R], scores[:R], rois_ext, masks[R:], scores[R:], o["scale"], im.shape)
            assert np.array_equal(boxes, ob)                                                     # a17
            om, obx = ohost.gpu_mask_voting(masks, boxes, scores, K, 100, im.shape[1], im.shape[0])    # a18 + a19
            assert [len(b) for b in got_b] == [len(b) for b in obx]
            assert np.array_equal(np.concatenate(got_b, 0), np.concatenate(obx, 0))
            assert np.array_equal(np.concatenate(got_m, 0), np.concatenate(om, 0), equal_nan=True)
            for name, (d, rel) in rep:
                lines.append("seed %d %-6s %-22s max|d|=%.3e rel=%.3e" % (seed, math, name, d, rel))
                assert rel < tol, lines[-1]
            # ---- free-running ---------------------------------------------------------------------------------------------
            st = _free_running(o, got_m, got_b, rois, rois_ext)
            stats.append(st)
            lines.append("seed %d %-6s free-running: rois within 0.01 px of an oracle roi %d/300 (%d at the same index), of those rois_ext "
                         "within 0.05 px %d; final instances device %d / oracle %d, matched (same class, IoU >= 0.9) %d; on matched: "
                         "|score| <= %.3e, |box| <= %.1f px, mask cells off by > 1e-3: %d of %d (max %.3e, mean %.3e)"
                         % (seed, math, st["same_rois"], st["same_order"], st["same_rois_ext"], st["n_dev"], st["n_orc"], st["matched"],
                            st["max_score"], st["max_box"], st["cells_off"], st["cells"], st["max_mask"], st["mean_mask"]))
    finally:
        nat.close()
        print("\n".join(lines))
        _log(lines)
    # loose floors (the recorded figures are the result; see the module docstring)
    assert np.mean([s["same_rois"] for s in stats]) >= FLOOR_ROIS[math], stats
    assert np.mean([s["matched"] / max(s["n_orc"], 1) for s in stats]) >= FLOOR_MATCHED[math], stats
    assert max(s["cells_off"] / max(s["cells"], 1) for s in stats) <= CEIL_CELLS_OFF[math], stats
    if math in CEIL_CELLS_REST:
        assert max(s["cells_off_rest"] for s in stats) <= CEIL_CELLS_REST[math], stats
