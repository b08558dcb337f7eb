"""CPU: the oracle (oracle/host.py + oracle/mnc_oracle.c) against the fixtures produced by the REFERENCE's own
code (tests/golden/make_golden.py).  Integer/index outputs bit-exact; float32 outputs bit-exact unless noted."""
import numpy as np
import pytest

import golden_inputs as GI
from oracle import host, native

SURVEY_ANCHORS = np.array([[-84, -40, 99, 55], [-176, -88, 191, 103], [-360, -184, 375, 199], [-56, -56, 71, 71],
                           [-120, -120, 135, 135], [-248, -248, 263, 263], [-36, -80, 51, 95], [-80, -168, 95, 183],
                           [-168, -344, 183, 359]], dtype=np.float64)


def test_anchors(golden):
    a = host.generate_anchors()
    assert np.array_equal(a, golden["anchors"])
    assert np.array_equal(a, SURVEY_ANCHORS)          # the computed table, not the comment


def test_bbox_transform(golden):
    pred = host.bbox_transform_inv(golden["bt_boxes"], golden["bt_deltas"])
    assert pred.dtype == np.float32 and np.array_equal(pred, golden["bt_pred"])
    clipped, inside = host.clip_boxes(pred, tuple(golden["bt_im_shape"]))
    assert np.array_equal(clipped, golden["bt_clipped"]) and np.array_equal(inside, golden["bt_inside"])
    assert np.array_equal(host.filter_small_boxes(clipped[:, :4], 16.0), golden["bt_small_keep"])
    assert host.bbox_transform_inv(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32)).shape == (0, 8)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_proposal_layer(golden, tag):
    fh, fw, seed = [int(v) for v in golden["prop_%s_meta" % tag]]
    pc = GI.proposal_case(fh, fw, seed)
    if tag == "small":
        assert np.array_equal(pc["cls_prob"], golden["prop_small_cls_prob"])   # generator stream is stable
        assert np.array_equal(pc["bbox_pred"], golden["prop_small_bbox_pred"])
    rois = host.proposal_forward(pc["cls_prob"], pc["bbox_pred"], pc["im_info"])
    want = golden["prop_%s_rois" % tag]
    assert rois.dtype == np.float32 and rois.shape == want.shape
    assert np.array_equal(rois, want)
    assert rois.shape[0] <= 300 and np.all(rois[:, 0] == 0)


def test_stage_bridge(golden):
    out = host.stage_bridge_forward_test(golden["sb_rois"], golden["sb_bbox_pred"], golden["sb_scores"],
                                         golden["sb_im_info"])
    assert np.array_equal(out, golden["sb_rois_ext"])


def test_mask_layer(golden):
    out = host.mask_layer_forward_test(golden["ml_in"])
    assert out.shape == (7, 1, 21, 21) and np.array_equal(out, golden["ml_out"])


@pytest.mark.parametrize("n,thr,seed", GI.NMS_CASES)
def test_nms_keep_bit_exact(golden, n, thr, seed):
    keep = native.gpu_nms(GI.nms_case(n, seed), thr)
    want = golden["nms_%d_%s_keep" % (n, str(thr).replace(".", "p"))]
    assert np.array_equal(np.array(keep, np.int64), want)


def test_nms_mask_matches_scan():
    dets = GI.nms_case(300, 77)
    order = dets[:, 4].argsort()[::-1]
    m = native.nms_mask(dets[order], 0.5)
    # lower-triangle words are never read by the scan (nms_kernel.cu:135) but are still produced (:39 commented out)
    assert m.shape == (300, 5)
    iu = np.triu_indices(5, 1)
    assert m[:64][:, 1:].any()
    for i in range(64):        # diagonal tile: only j > i
        assert int(m[i, 0]) & ((1 << (i + 1)) - 1) == 0


@pytest.mark.parametrize("tag", ["small", "full"])
def test_gpu_mask_voting(golden, tag):
    n, H, W, seed = GI.VOTING_CASES[tag]
    vc = GI.voting_case(n, H, W, seed)
    lm, lb = host.gpu_mask_voting(vc["masks"], vc["boxes"], vc["scores"], 21, 100, W, H)
    assert np.array_equal(np.array([len(b) for b in lb]), golden["vote_%s_count" % tag])
    box, mask = np.concatenate(lb, 0), np.concatenate(lm, 0)
    assert box.dtype == np.float64                      # int32 hstack float32 -> float64 (mask_transform.py:276)
    assert np.array_equal(box, golden["vote_%s_box" % tag])
    assert np.array_equal(mask, golden["vote_%s_mask" % tag])
    assert 0 < box.shape[0] <= 100 + 20


def test_mv_direct(golden):
    mc = GI.mv_case(8)
    rm, rb = native.mv(mc["boxes"], mc["masks"], mc["inds"], mc["start"], mc["weights"], mc["H"], mc["W"])
    assert np.array_equal(rb, golden["mv_box"]) and np.array_equal(rm, golden["mv_mask"])
    # result 3 never reaches 0.4 -> both bounds default to W/2, H/2 (mv_kernel.cu:149,173)
    assert list(rb[3]) == [mc["W"] // 2, mc["H"] // 2, mc["W"] // 2, mc["H"] // 2]
    # R == 0 returns empty arrays instead of the reference's IndexError
    em, eb = native.mv(mc["boxes"], mc["masks"], np.zeros(0, np.int32), np.zeros(0, np.int32),
                       np.zeros(0, np.float32), mc["H"], mc["W"])
    assert em.shape == (0, 1, 21, 21) and eb.shape == (0, 4)


def test_im_detect(golden):
    dc = GI.detect_case(9)
    data, im_info, scale = host.prepare_mnc_args(dc["im"])
    assert scale == 1.0 and data.shape == (1, 3, 600, 1000)
    assert np.array_equal(im_info, golden["det_im_info"])
    assert np.array_equal(data[0, :, :4, :4], golden["det_data_corner"])
    assert abs(data.astype(np.float64).sum() - golden["det_data_sum"][0]) < 1e-6 * abs(golden["det_data_sum"][0])
    b = dc["blobs"]
    boxes, masks, scores = host.im_detect_tail(b["rois"], b["mask_proposal"], b["seg_cls_prob"], b["rois_ext"],
                                               b["mask_proposal_ext"], b["seg_cls_prob_ext"], scale, dc["im"].shape)
    # the reference under numpy>=2 promotes the un-scaling to float64; at scale 1.0 the values are identical
    assert np.array_equal(boxes.astype(np.float64), golden["det_boxes"].astype(np.float64))
    assert tuple(golden["det_masks_shape"]) == masks.shape and np.array_equal(scores, golden["det_scores"])


def test_bbox_overlaps_against_pyx_transcription():
    rng = np.random.default_rng(3)
    a = GI._boxes(rng, 50, 300, 200).astype(np.float64)
    q = GI._boxes(rng, 7, 300, 200).astype(np.float64)
    got = native.bbox_overlaps(a, q)
    want = np.zeros((50, 7))
    for k in range(7):
        qa = (q[k, 2] - q[k, 0] + 1) * (q[k, 3] - q[k, 1] + 1)
        for n in range(50):
            iw = min(a[n, 2], q[k, 2]) - max(a[n, 0], q[k, 0]) + 1
            ih = min(a[n, 3], q[k, 3]) - max(a[n, 1], q[k, 1]) + 1
            if iw > 0 and ih > 0:
                want[n, k] = iw * ih / ((a[n, 2] - a[n, 0] + 1) * (a[n, 3] - a[n, 1] + 1) + qa - iw * ih)
    assert np.array_equal(got, want)


def test_resize_identity_and_shape():
    im = np.random.default_rng(0).uniform(0, 255, (30, 50, 3)).astype(np.float32)
    assert np.array_equal(host.resize_bilinear_cv(im, 1.0, 1.0), im)
    up = host.resize_bilinear_cv(im, 1.6, 1.6)
    assert up.shape == (48, 80, 3) and up.min() >= im.min() - 1e-3 and up.max() <= im.max() + 1e-3
    # scale cap: 375x500 -> 600/375 = 1.6 (round(1.6*500)=800 <= 1000); 333x1000 -> capped at 1.0
    _, s = host.prep_im_for_blob(np.zeros((375, 500, 3), np.uint8))
    assert s == 1.6
    _, s = host.prep_im_for_blob(np.zeros((333, 1000, 3), np.uint8))
    assert s == 1.0
