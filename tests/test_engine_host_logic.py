"""CPU: the HOST logic of the executor and of the reference-shaped modules (graph plan, fusions, Concat views, layouts,
Python-layer plumbing, demo.im_detect, gpu_mask_voting orchestration) with the C ABI replaced by the test double in
tests/fake_backend.py.  Kernel parity is the job of the -m gpu tests; this file guarantees that what reaches the
kernels is shaped and ordered correctly, by comparing every blob with the oracle graph."""
import numpy as np
import pytest

import fake_backend
import mnc_amd
from mnc_amd import models, synth
from oracle import host as ohost
from oracle import net as onet

mnc_amd.install_paths()

BLOBS = ["conv1_1", "pool1", "conv3_3", "conv5_3", "rpn_cls_prob_reshape", "rpn_bbox_pred", "rois",
         "roi_interpolate_conv5", "mask_output", "mask_proposal", "mask_proposal_resize", "roi_interpolate_conv5_box",
         "roi_interpolate_conv5_mask", "fc6", "fc7", "fc7_mask", "join_box_mask", "cls_prob", "seg_cls_prob",
         "bbox_pred", "rois_ext", "roi_interpolate_conv5_ext", "mask_proposal_ext", "seg_cls_prob_ext",
         "bbox_pred_ext", "cls_prob_ext"]


@pytest.fixture()
def fake_gpu(monkeypatch):
    import gc
    yield fake_backend.install(monkeypatch)
    gc.collect()          # Nets left behind by a failing test must be finalised while the double is still installed


def _inputs(H, W, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-120, 130, (1, 3, H, W)).astype(np.float32), np.array([[H, W, 1.0]], np.float32)


@pytest.mark.parametrize("fuse,math", [(True, "fp32"), (False, "fp32"), (True, "bf16x3"), (True, "f16")])
def test_reduced_graph_every_blob(fake_gpu, monkeypatch, fuse, math):
    from mnc_amd import engine
    from mnc_amd.engine import Net
    monkeypatch.setattr(engine, "_X3_MIN_FLOPS", 0.0)      # bf16x3: every InnerProduct takes the split-weight path
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    net = Net(path, w, 1, device_id=0, fuse=fuse, math=math)
    data, im_info = _inputs(96, 160, 0)
    net.blobs["data"].reshape(*data.shape)
    net.blobs["im_info"].reshape(*im_info.shape)
    out = net.forward(data=data, im_info=im_info)
    assert set(out) == {"cls_prob", "cls_prob_ext", "seg_cls_prob_ext", "bbox_pred_ext"}
    calls = fake_gpu.calls
    if math == "fp32":
        assert not any(k.endswith(("_sm", "_pre")) for k in calls)
        assert (calls.get("mnc_box_mask_pool_ex", 0) > 0) == fuse and ("mnc_mask_pool" in calls) == (not fuse)
    else:
        # reduced-precision InnerProducts: the per-RoI producers write the rows in the GEMM's own stage-major 2-byte form and the
        # six big InnerProducts on per-RoI features (fc6_maskest, fc6, fc6_mask of both stages) take them without converting
        ex = "mnc_fc_f16_ex" if math == "f16" else "mnc_fc_bf16x3_ex"
        # (twice when fewer proposals survive than were speculated and the heads are re-run on the exact count)
        runs = calls.get("mnc_roi_warp_sm", 0) // 2
        # every reduced-precision InnerProduct goes through the general entry point: 6 per stage here (fc6_maskest, mask_pred, fc6,
        # fc7, fc6_mask, fc7_mask -- _X3_MIN_FLOPS = 0 makes mask_pred one of them, and with K = 32 it takes the split-bf16 kernel
        # in the f16 mode too; the merged sibling heads stay fp32); none converts its input rows itself
        n_ex = calls.get("mnc_fc_f16_ex", 0) + calls.get("mnc_fc_bf16x3_ex", 0)
        # round 6: fc6 + fc6_mask and fc7 + fc7_mask of a stage are one mnc_fc_lowp_pair call each (fp16 and split bf16) (their inputs both
        # exist when the first runs: the one-pass pooling of the fused plan)
        n_pair = calls.get("mnc_fc_lowp_pair", 0)
        assert runs in (1, 2) and n_ex + 2 * n_pair == 12 * runs and "mnc_roi_warp" not in calls
        assert n_pair == (4 * runs if fuse else 0)
        assert calls.get(ex, 0) + 2 * n_pair >= 10 * runs
        assert "mnc_fc_f16" not in calls and "mnc_fc_bf16x3" not in calls
        # the box-feature Pooling and MaskPooling + Pooling of a stage read the same tensor: one pass, both second outputs
        assert calls.get("mnc_box_mask_pool_ex") == 2 * runs and "mnc_mask_pool_sm" not in calls and "mnc_maxpool2_rhwc_sm" not in calls
    if math == "bf16x3":
        # round 6: the split-bf16 trunk keeps its packed form between the MFMA layers too ('c8x'), conv5_3 fp32 for the RoI layers
        # (fused plan: the four pooled layers run as conv + ReLU + pool in one launch, mnc_conv3x3_lowp_pool)
        assert calls.get("mnc_conv3x3_bf16x3_pk") == (8 if fuse else 12) and calls.get("mnc_conv3x3_c3_fmt") == 1
        assert calls.get("mnc_maxpool2_c8_bf16x3", 0) == (0 if fuse else 4) and calls.get("mnc_conv3x3_lowp_pool", 0) == (4 if fuse else 0)
        assert net.blobs["conv5_2"].layout == "c8x" and net.blobs["conv5_3"].layout == "c8"
    if math == "f16":
        # 2-byte trunk activations: conv1_1 .. conv5_2 write packed fp16, conv5_3 (read by the RoI layers) fp32
        assert calls.get("mnc_conv3x3_f16_pk") == (8 if fuse else 12) and calls.get("mnc_conv3x3_c3_fmt") == 1
        assert calls.get("mnc_maxpool2_c8_f16", 0) == (0 if fuse else 4) and calls.get("mnc_conv3x3_lowp_pool", 0) == (4 if fuse else 0)
        assert net.blobs["conv5_2"].layout == "c8h" and net.blobs["conv5_3"].layout == "c8"
    ref = onet.forward(w, data, im_info)
    names = BLOBS + ([] if fuse else ["roi_interpolate_conv5_premax", "roi_mask_conv5", "roi_mask_conv5_ext"])
    if fuse:
        # the convolution applies the following MAX 2x2/2 pool in its epilogue (fp32 Winograd; round 6: the reduced-precision kernel too): conv3_3 is not materialised
        assert not (net.blobs["conv3_3"]._dev_valid or net.blobs["conv3_3"]._host_valid)
        names = [("pool3" if n == "conv3_3" else n) for n in names]
    # split weights are exact to 2^-16 only; 1e-3 is the end-to-end bar.  f16 rounds both FC operands to 11 bits: 1e-2 here
    # (plumbing check; the measured accuracy of that mode is reported by the GPU tests)
    tol = {"fp32": 1e-4, "bf16x3": 1e-3, "f16": 1e-2}[math]
    if math == "f16":
        # fp16 rounding in the trunk can flip a borderline NMS / min-size decision, after which the RoI lists differ row by
        # row: only the blobs upstream of the ProposalLayer are compared here (the GPU tests teacher-force the rest)
        names = ["conv1_1", "pool1", "pool3" if fuse else "conv3_3", "conv5_3", "rpn_cls_prob_reshape", "rpn_bbox_pred"]
        assert net.blobs["seg_cls_prob_ext"].data.shape[1] == 21 and net.blobs["rois"].data.shape[1] == 5
    for n in names:
        got, want = net.blobs[n].data, ref[n]
        assert got.shape == want.shape, n
        assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-6), n
    if math == "f16":
        net.close()
        return
    # a second image of another size through the same net (buffers are re-used / re-grown, shapes are dynamic)
    data2, info2 = _inputs(130, 203, 1)
    net.forward(data=data2, im_info=info2)
    ref2 = onet.forward(w, data2, info2)
    for n in ("conv5_3", "rois", "seg_cls_prob", "mask_proposal_ext", "seg_cls_prob_ext"):
        assert net.blobs[n].data.shape == ref2[n].shape, n
        assert np.abs(net.blobs[n].data - ref2[n]).max() <= tol * max(np.abs(ref2[n]).max(), 1e-6), n
    assert sorted(net.params["fc6"][0].shape) == sorted(w["fc6"][0].shape)
    net.close()


def test_fusion_plan(fake_gpu):
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    net = Net(path, w, 1, device_id=0)
    L = {l.name: l for l in net._layers}
    assert L["relu3_2"].skip and L["conv3_2"].relu
    assert L["roi_interpolate_conv5_premax"].fused_pool and L["roi_interpolate_conv5"].skip
    assert not L["roi_interpolate_conv5_ext"].fused_pool           # stage 4 warps 14x14 directly, its consumers differ
    assert L["mask_pred"].act == 2 and L["mask_output"].skip
    assert L["mask_pooling"].fused_pool and L["roi_interpolate_conv5_mask"].skip
    assert not L["roi_interpolate_conv5_box"].skip                  # its bottom is shared with other consumers
    assert net.blobs["fc7"]._view is not None and net.blobs["fc7_mask"]._view[1] == 0 and net.blobs["fc7"]._view[1] == 512
    assert net.outputs == ["cls_prob", "cls_prob_ext", "seg_cls_prob_ext", "bbox_pred_ext"]
    assert [m.name for m in L["cls_score"].group] == ["cls_score", "seg_cls_score", "bbox_pred"]
    assert L["bbox_pred"].group_leader is L["cls_score"] and net.blobs["bbox_pred"]._view[1] == 42
    # the box and the mask branch's InnerProducts of a stage are launched in pairs (mnc_fc_pair; the reduced-precision modes: mnc_fc_lowp_pair)
    for a, b in (("fc6", "fc6_mask"), ("fc7", "fc7_mask"), ("fc6_ext", "fc6_mask_ext"), ("fc7_ext", "fc7_mask_ext")):
        assert L[a].pair is L[b] and L[b].pair_leader is L[a], (a, b)
    assert L["fc6_maskest"].pair is None and L["fc6_maskest"].pair_leader is None
    import demo
    im = np.random.default_rng(0).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    fake_gpu.calls.clear()
    demo.im_detect(im, net)
    stages = fake_gpu.calls.get("mnc_box_mask_pool_ex")              # head stages run (two; four when the heads are re-run on the exact RoI count)
    assert stages in (2, 4)
    assert fake_gpu.calls.get("mnc_fc_pair") == 2 * stages          # fc6 + fc6_mask, fc7 + fc7_mask per stage
    assert fake_gpu.calls.get("mnc_fc") == 3 * stages               # fc6_maskest, mask_pred, the sibling classifiers' GEMM per stage
    net.close()
    net3 = Net(path, w, 1, device_id=0, math="bf16x3")           # round 6: the reduced-precision modes pair the same layers (mnc_fc_lowp_pair)
    L3 = {l.name: l for l in net3._layers}
    assert L3["fc6"].pair is L3["fc6_mask"] and L3["fc7"].pair is L3["fc7_mask"] and L3["fc6_maskest"].pair is None
    net3.close()


def test_demo_im_detect_and_voting(fake_gpu):
    import demo
    from mnc_amd.engine import Net
    from transform.mask_transform import gpu_mask_voting
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=2)
    net = Net(path, w, 1, device_id=0)
    im = np.random.default_rng(4).integers(0, 256, (75, 100, 3), dtype=np.uint8)     # -> 600x800, scale 8.0
    boxes, masks, scores = demo.im_detect(im, net)
    oboxes, omasks, oscores = onet.im_detect(w, im)
    assert boxes.shape == oboxes.shape and boxes.dtype == oboxes.dtype == np.float32
    assert np.abs(boxes - oboxes).max() < 1e-3 and np.abs(scores - oscores).max() < 1e-4
    assert np.abs(masks - omasks).max() < 1e-4
    lm, lb = gpu_mask_voting(masks, boxes, scores, 21, 100, im.shape[1], im.shape[0])
    om, ob = ohost.gpu_mask_voting(masks, boxes, scores, 21, 100, im.shape[1], im.shape[0])
    assert np.array_equal(np.concatenate(lb, 0), np.concatenate(ob, 0))
    assert np.array_equal(np.concatenate(lm, 0), np.concatenate(om, 0))
    # results alias device buffers that the next image reuses: an array that was copied to the host keeps its values, one that
    # was never looked at refuses to hand out the next image's data
    b2, m2, s2 = demo.im_detect(im, net)
    kept = np.asarray(b2).copy()
    b3, m3, s3 = demo.im_detect(im[::-1].copy(), net)
    assert np.array_equal(np.asarray(b2), kept) and b3.is_current() and not m2.is_current()
    with pytest.raises(RuntimeError, match="reused by a later image"):
        np.asarray(m2)
    net.close()


def test_instance_block_and_gatherer_host_logic(fake_gpu):
    """The host side of the device-resident result path (InstanceBlock head/records layout, the second copy when more rows
    than max_per_image tie at the threshold, InstanceGatherer's device transport) against the oracle, on the test double.
    The same function runs on the real kernels and a real RCCL communicator in tests/test_gpu_engine.py."""
    import test_gpu_engine as T
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    net = Net(path, w, 1, device_id=0)
    try:
        T.test_device_instance_block_and_rccl_gather((net, w))
    finally:
        net.close()


def test_pylayers_against_reference_fixtures(fake_gpu, golden):
    """The product's Python layers (mnc_amd/lib/pylayer) driven through caffe.Layer's protocol, against the fixtures
    produced by the reference's own layers."""
    import golden_inputs as GI
    from pylayer.mask_layer import MaskLayer
    from pylayer.proposal_layer import ProposalLayer
    from pylayer.stage_bridge_layer import StageBridgeLayer

    class B(object):
        def __init__(self, a=None):
            self.data = np.zeros((1,), np.float32) if a is None else np.ascontiguousarray(a, np.float32)

        def reshape(self, *d):
            if tuple(d) != self.data.shape:
                self.data = np.zeros(d, np.float32)

    def run(cls, bottoms, param_str=""):
        layer = cls()
        layer.param_str_, layer.phase = param_str, "TEST"
        bottom, top = [B(b) for b in bottoms], [B()]
        layer.setup(bottom, top)
        layer.reshape(bottom, top)
        layer.forward(bottom, top)
        return top[0].data

    for tag in ("small", "full"):
        fh, fw, seed = [int(v) for v in golden["prop_%s_meta" % tag]]
        pc = GI.proposal_case(fh, fw, seed)
        rois = run(ProposalLayer, [pc["cls_prob"], pc["bbox_pred"], pc["im_info"]], "{'feat_stride': 16}")
        assert np.array_equal(rois, golden["prop_%s_rois" % tag])
    out = run(StageBridgeLayer, [golden["sb_rois"], golden["sb_bbox_pred"], golden["sb_scores"], golden["sb_im_info"]])
    assert np.array_equal(out, golden["sb_rois_ext"])
    assert np.array_equal(run(MaskLayer, [golden["ml_in"]]), golden["ml_out"])


def test_voting_host_code_against_reference_fixture(fake_gpu, golden):
    import golden_inputs as GI
    from transform.mask_transform import gpu_mask_voting
    n, H, W, seed = GI.VOTING_CASES["small"]
    vc = GI.voting_case(n, H, W, seed)
    lm, lb = gpu_mask_voting(vc["masks"], vc["boxes"], vc["scores"], 21, 100, W, H)
    assert np.array_equal(np.concatenate(lb, 0), golden["vote_small_box"])
    assert np.array_equal(np.concatenate(lm, 0), golden["vote_small_mask"])


def test_unsupported_layer_geometry_is_refused(fake_gpu, tmp_path):
    """Convolution fields the kernels do not implement (dilation, group, kernel_h / pad_w ...) and odd per-RoI pooling sizes raise
    instead of running with the wrong geometry."""
    from mnc_amd.engine import Net
    base = """name: "t"
input: "data"
input_shape { dim: 1 dim: 3 dim: 32 dim: 32 }
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 32 kernel_size: 3 pad: 1 %s } }
"""
    w = {"c1": [np.zeros((32, 3, 3, 3), np.float32), np.zeros(32, np.float32)]}
    for extra, word in (("dilation: 2", "dilation"), ("group: 2", "group"), ("kernel_h: 3", "kernel_h"), ("pad_w: 1", "pad_w")):
        path = tmp_path / ("%s.prototxt" % word)
        path.write_text(base % extra)
        with pytest.raises(NotImplementedError, match=word):
            Net(str(path), w, 1, device_id=0)
    ok = tmp_path / "ok.prototxt"
    ok.write_text(base % "stride: 1")
    Net(str(ok), w, 1, device_id=0).close()


def test_trained_weights_into_unpinned_layers_warn_loudly(fake_gpu, tmp_path, monkeypatch):
    """A real .caffemodel / .h5 loaded into a graph with ROIWarping / MaskResize / MaskPooling (source in the absent caffe-mnc
    submodule, semantics from oracle/SPEC.md) produces a PARITY UNPINNED warning unless acknowledged; dict / .npz weights and
    graphs without those layers do not."""
    import warnings
    from mnc_amd import caffemodel
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    monkeypatch.setattr(caffemodel, "load_weights", lambda p: w)
    with pytest.warns(UserWarning, match="PARITY UNPINNED.*MaskPooling, MaskResize, ROIWarping"):
        Net(path, str(tmp_path / "mnc_model.caffemodel.h5"), 1, device_id=0).close()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        Net(path, w, 1, device_id=0).close()
        Net(path, str(tmp_path / "w.npz"), 1, device_id=0).close()
        monkeypatch.setenv("MNC_ACCEPT_UNPINNED_LAYERS", "1")
        Net(path, str(tmp_path / "mnc_model.caffemodel.h5"), 1, device_id=0).close()


def test_blob_semantics(fake_gpu):
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    net = Net(path, synth.synthetic_weights(path, seed=1), 1, device_id=0)
    b = net.blobs["data"]
    b.reshape(1, 3, 8, 8)
    assert b.data.shape == (1, 3, 8, 8) and b.count == 192 and b.channels == 3
    with pytest.raises(KeyError):
        net.forward(nonexistent=np.zeros(3, np.float32))
    assert not net.blobs["conv5_3"].data.any()          # never produced: zero-filled, like a fresh caffe blob
    net.close()


def test_config_surface(tmp_path):
    from mnc_config import cfg, cfg_from_file
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 300 and cfg.TEST.RPN_NMS_THRESH == 0.7 and cfg.MASK_SIZE == 21
    p = tmp_path / "c.yml"
    p.write_text("EXP_DIR: mnc_5stage\nMASK_SIZE: 21\nTRAIN:\n  RPN_POST_NMS_TOP_N: 300\n  IMS_PER_BATCH: 1\n"
                 "  BBOX_NORMALIZE_TARGETS_PRECOMPUTED: True\n")
    cfg_from_file(str(p))
    assert cfg.EXP_DIR == "mnc_5stage" and cfg.TRAIN.RPN_POST_NMS_TOP_N == 300
    p.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        cfg_from_file(str(p))
    p.write_text("MASK_SIZE: 'x'\n")
    with pytest.raises(ValueError):
        cfg_from_file(str(p))
    cfg.EXP_DIR = "default"
    cfg.TRAIN.RPN_POST_NMS_TOP_N = 2000
    cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED = False


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/experiments/cfgs"), reason="reference checkout not mounted")
def test_config_defaults_and_experiment_files_of_the_reference(monkeypatch):
    """Every `__C.<KEY> = <literal>` default of the reference's lib/mnc_config.py is declared here with the same value, and
    its three experiment files (experiments/cfgs/VGG16/*.yml) merge without unknown-key errors."""
    import ast
    import copy
    import glob
    import re
    import mnc_config
    seen = 0
    for line in open("/root/reference/lib/mnc_config.py"):
        m = re.match(r"^__C\.([A-Z_.]+) *= *(.+?)\s*(#.*)?$", line)
        if not m:
            continue
        try:
            want = ast.literal_eval(m.group(2))
        except (ValueError, SyntaxError):
            continue                                  # edict(), np.array(...), os.path expressions
        node = mnc_config.cfg
        for part in m.group(1).split("."):
            assert part in node, m.group(1)
            node = node[part]
        same = node == want or (isinstance(want, (list, tuple)) and tuple(node) == tuple(want))
        assert same, (m.group(1), node, want)
        seen += 1
    assert seen >= 60
    saved = copy.deepcopy(dict(mnc_config.cfg))
    try:
        for f in sorted(glob.glob("/root/reference/experiments/cfgs/VGG16/*.yml")):
            mnc_config.cfg_from_file(f)
        assert mnc_config.cfg.EXP_DIR == "mnc_5stage" and mnc_config.cfg.TRAIN.RPN_POST_NMS_TOP_N == 300     # last file wins
        mnc_config.cfg_from_file("/root/reference/experiments/cfgs/VGG16/cfm.yml")
        assert list(mnc_config.cfg.TEST.MAX_ROIS_GPU) == [2000, 500] and mnc_config.cfg.TEST.GROUP_SCALE == 3
    finally:
        for k, v in saved.items():
            mnc_config.cfg[k] = v


def test_speculative_roi_count_equals_synchronous_path(fake_gpu, monkeypatch):
    """The heads are launched on RPN_POST_NMS_TOP_N rows before the RoI count is known and re-run on the exact count when
    fewer proposals survive (a small image: far fewer than 300): same blobs as with the mid-forward read-back."""
    from mnc_amd.engine import Net
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    data, im_info = _inputs(64, 96, 4)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MNC_SPECULATE_ROIS", flag)
        net = Net(path, w, 1, device_id=0)
        assert net._speculate == (flag == "1")
        net.forward(data=data, im_info=im_info)
        outs[flag] = {n: net.blobs[n].data.copy() for n in ("rois", "seg_cls_prob", "rois_ext", "mask_proposal_ext", "cls_prob_ext")}
        net.close()
    assert 0 < outs["1"]["rois"].shape[0] < 300
    for n in outs["1"]:
        assert outs["1"][n].shape == outs["0"][n].shape and np.array_equal(outs["1"][n], outs["0"][n]), n


def _cfm_inputs(seed, N, H, W, R):
    """A batch of N pyramid levels (smaller levels zero-padded, utils/blob.py:im_list_to_blob), R rois over them (some
    degenerate / partly outside, to reach the empty-bin and clipping branches) and binary 14x14 masks."""
    rng = np.random.default_rng(seed)
    data = np.zeros((N, 3, H, W), np.float32)
    for n in range(N):
        h, w = H - 9 * n, W - 14 * n
        data[n, :, :h, :w] = rng.uniform(-120, 130, (3, h, w))
    x1, y1 = rng.uniform(0, W - 20, R), rng.uniform(0, H - 20, R)
    bw, bh = rng.uniform(4, W * 0.8, R), rng.uniform(4, H * 0.8, R)
    rois = np.stack([rng.integers(0, N, R).astype(np.float64), x1, y1, np.minimum(x1 + bw, W + 30), np.minimum(y1 + bh, H + 30)],
                    1).astype(np.float32)
    rois[0, 1:] = [5, 5, 5, 5]                       # one-pixel roi: 1x1 map cell repeated over all bins
    rois[1, 1:] = [W + 40, H + 40, W + 90, H + 90]     # entirely outside: every bin empty -> zeros
    masks = (rng.uniform(0, 1, (R, 1, 14, 14)) >= 0.4).astype(np.float32)
    return data, rois, masks


@pytest.mark.parametrize("fuse", [True, False])
def test_cfm_graph_every_blob(fake_gpu, fuse):
    """CFM test graph: image batch > 1 through the trunk, ROIPooling with batch indices, binary MaskPooling."""
    from mnc_amd.engine import Net
    path = models.write_cfm_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=4)
    net = Net(path, w, 1, device_id=0, fuse=fuse)
    for seed, (N, H, W, R) in enumerate([(3, 96, 144, 37), (1, 70, 81, 5), (2, 64, 64, 0)]):
        data, rois, masks = _cfm_inputs(seed, N, H, W, max(R, 2))
        rois, masks = rois[:R], masks[:R]
        net.blobs["data"].reshape(*data.shape)
        net.blobs["rois"].reshape(*rois.shape)
        net.blobs["masks"].reshape(*masks.shape)
        out = net.forward(data=data, rois=rois, masks=masks)
        assert set(out) == {"mask_prob", "cls_prob", "seg_cls_prob", "bbox_pred"}
        ref = onet.forward_cfm(w, data, rois, masks)
        names = ["conv1_1", "pool2", "conv5_3", "roi_pooling_conv5", "roi_pooling_conv5_mask", "roi_mask_conv5_pool", "fc7",
                 "fc7_mask", "fc6_maskest", "join_box_mask", "mask_prob", "cls_prob", "seg_cls_prob", "bbox_pred"]
        for n in names + ([] if fuse else ["roi_mask_conv5", "mask_pred"]):
            got = net.blobs[n].data if n not in out else out[n]
            want = ref[n] if n != "mask_pred" else None
            if want is None:
                continue
            assert got.shape == want.shape, n
            if want.size:
                assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6), n
    net.close()


def test_device_image_prep_plumbing(fake_gpu):
    """net.prep_image (tap tables, level offsets, padding, DeviceArray hand-over to forward) equals the numpy functions of
    lib/utils/blob.py bit for bit -- for the MNC single-scale input and for a CFM pyramid group."""
    from mnc_amd.engine import Net
    from mnc_config import cfg
    from utils.blob import (im_list_to_blob, prep_im_for_blob, prep_im_for_blob_cfm, prep_im_for_blob_cfm_device,
                            prep_im_for_blob_device)
    path = models.write_cfm_test_prototxt(width_div=8)
    net = Net(path, synth.synthetic_weights(path, seed=4), 1, device_id=0)
    rng = np.random.default_rng(9)
    for H, W in [(75, 100), (60, 60), (90, 37)]:
        im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        host, hs = prep_im_for_blob(im, cfg.PIXEL_MEANS, 120, 170)
        dev, ds = prep_im_for_blob_device(net, im, cfg.PIXEL_MEANS, 120, 170)
        assert ds == hs and dev.shape == (1,) + host.transpose(2, 0, 1).shape
        assert np.array_equal(np.asarray(dev), im_list_to_blob([host]))
        old = cfg.TEST.MAX_SIZE
        cfg.TEST.MAX_SIZE = 260
        try:
            hb, hf = prep_im_for_blob_cfm(im, [100, 150, 222])
            db, df = prep_im_for_blob_cfm_device(net, im, [100, 150, 222])
        finally:
            cfg.TEST.MAX_SIZE = old
        assert np.array_equal(hf, df) and db.shape == hb.shape and np.array_equal(np.asarray(db), hb)
    # identity scale: the general path returns the mean-subtracted pixels themselves
    im = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    dev = net.prep_image(im, cfg.PIXEL_MEANS, [1.0])
    f = im.astype(np.float32, copy=True)
    f -= cfg.PIXEL_MEANS
    assert np.array_equal(np.asarray(dev)[0], f.transpose(2, 0, 1))
    # forward adopts the DeviceArray: same conv1_1 as with the host blob
    rois = np.array([[0, 4, 4, 60, 50]], np.float32)
    masks = np.ones((1, 1, 14, 14), np.float32)
    net.forward(data=dev, rois=rois, masks=masks)
    a = net.blobs["conv1_1"].data.copy()
    net.forward(data=np.asarray(dev).copy(), rois=rois, masks=masks)
    assert np.array_equal(a, net.blobs["conv1_1"].data)
    with pytest.raises(TypeError):
        net.prep_image(im.astype(np.float32), cfg.PIXEL_MEANS, [1.0])
    net.close()


RESNET_BLOBS = ["conv1", "pool1", "res2a_branch1", "res2a_branch2a", "res2a_branch2b", "res2a", "res2c", "res3a", "res3d", "res4a",
                "res4f", "rpn_cls_prob_reshape", "rpn_bbox_pred", "rois", "roi_interpolate_conv5", "mask_proposal", "fc7",
                "seg_cls_prob", "bbox_pred", "rois_ext", "mask_proposal_ext", "seg_cls_prob_ext"]


def test_resnet50_graph_f16_mode_plumbing(fake_gpu):
    """f16 math mode on the ResNet trunk: every convolution family takes its fp16 variant; trunk blobs within fp16's reach."""
    from mnc_amd.engine import Net
    path = models.write_mnc_resnet50_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=5)
    net = Net(path, w, 1, device_id=0, math="f16")
    data, im_info = _inputs(96, 160, 0)
    net.forward(data=data, im_info=im_info)
    # 2-byte activations between the layers that can take them: the stem, MAX 3x3/2, every 1x1 convolution whose input has a
    # multiple of 16 channels (at this width: all but the 8-channel bottlenecks of res2) and the tuned 3x3 kernels; res4f is
    # written as fp32 for the RoI layers
    by_name = {L.name: L for L in net._layers}
    assert by_name["conv1"].out_h and by_name["res4a_branch2a"].out_h and by_name["res4c_branch2b"].out_h
    assert by_name["res4e_branch2c"].out_h and not by_name["res4f_branch2c"].out_h and not by_name["rpn_conv_3x3"].out_h
    assert net._conv_fast1x1(by_name["res3a_branch1"]) and not net._conv_fast1x1(by_name["res2a_branch2c"])     # Cin = 8
    calls = fake_gpu.calls
    assert calls.get("mnc_conv_stem_c3_fmt") == 1 and calls.get("mnc_maxpool_c8_f16") == 1 and calls.get("mnc_conv1x1_f16_pk", 0) >= 20
    assert net.blobs["res4e"].layout == "c8h" and net.blobs["res4f"].layout == "c8"
    ref = {}
    onet.trunk_resnet50(w, data, ref)
    for n in ("conv1", "res2a", "res3d", "res4f"):
        got, want = net.blobs[n].data, ref[n]
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-2 * np.abs(want).max(), n
    net.close()


@pytest.mark.parametrize("fuse", [True, False])
def test_resnet50_graph_every_blob(fake_gpu, fuse):
    """the 5-stage cascade on a ResNet-50 C4 trunk -- stem conv, BatchNorm/Scale folded into the convolutions,
    strided 1x1 convolutions, MAX 3x3/2, residual adds (folded into branch2c's epilogue when fusing) -- blob by blob against the
    unfolded oracle graph."""
    from mnc_amd.engine import Net
    path = models.write_mnc_resnet50_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=5)
    net = Net(path, w, 1, device_id=0, fuse=fuse)
    kinds = {L.name: net._conv_kind(L) for L in net._layers if L.type == "Convolution"}
    assert kinds["conv1"] == "stem" and kinds["res2a_branch2a"] == "general" and kinds["rpn_conv_3x3"] == "fast3x3"
    assert kinds["rpn_cls_score"] == "nchw1x1" and kinds["res3a_branch1"] == "general"
    folded = [L for L in net._layers if L.type == "Convolution" and L.residual]
    assert len(folded) == (13 if fuse else 0)          # 3 + 4 + 6 bottleneck blocks in C4
    assert all(L.skip for L in net._layers if L.type in ("BatchNorm", "Scale"))
    for seed, (H, W) in enumerate([(96, 160), (131, 203)]):
        data, im_info = _inputs(H, W, seed)
        net.blobs["data"].reshape(*data.shape)
        net.forward(data=data, im_info=im_info)
        ref = onet.forward_resnet50(w, data, im_info)
        for n in RESNET_BLOBS:
            b = net.blobs[n]
            if not (b._dev_valid or b._host_valid):          # branch2c / shortcut blobs consumed inside a folded epilogue
                continue
            got, want = b.data, ref[n]
            assert got.shape == want.shape, n
            assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-6), n
    assert [p.data.shape for p in net.params["bn_conv1"]] == [(16,), (16,), (1,)] and len(net.params["res2a_branch2a"]) == 1
    net.close()


def test_detect_image_equals_the_demo_body(fake_gpu, monkeypatch):
    """Net.detect_image (one launch sequence per image, results through pinned memory) == demo.im_detect + gpu_mask_voting, for an
    image whose proposals fill RPN_POST_NMS_TOP_N and for one where fewer survive (the exact-count re-run).  Direct launches here;
    capture / replay of the same sequence is the GPU suite's (tests/test_gpu_engine.py)."""
    import demo
    from mnc_amd import models, synth
    from mnc_amd.engine import Net
    from mnc_amd.instances import split_records
    from mnc_config import cfg
    from transform.mask_transform import gpu_mask_voting
    path = models.write_mnc_5stage_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=1)
    for post in (300, 12):
        monkeypatch.setitem(cfg.TEST, "RPN_POST_NMS_TOP_N", post)
        net = Net(path, w, 1)
        ref = Net(path, w, 1)
        try:
            for seed in (0, 1):
                im = np.random.default_rng(seed).integers(0, 256, (48, 64, 3), dtype=np.uint8)
                counts, rec = net.detect_image(im, use_graph=False)
                b, m, s = demo.im_detect(im, ref)
                lm, lb = gpu_mask_voting(m, b, s, 21, 100, im.shape[1], im.shape[0])
                gm, gb = split_records(rec, counts[1:], 21)
                assert [len(x) for x in gb] == [len(x) for x in lb]
                assert np.array_equal(np.concatenate(gb, 0), np.concatenate(lb, 0))
                assert np.array_equal(np.concatenate(gm, 0), np.concatenate(lm, 0), equal_nan=True)
        finally:
            net.close()
            ref.close()
