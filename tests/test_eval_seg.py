"""CPU: the caller on the output side of the hot path -- `tools/test_net.py --task seg`:
TesterWrapper's per-image loop and the SDS mAP^r evaluation -- against outputs of the REFERENCE'S OWN CODE on a synthetic
VOCdevkitSDS (tests/golden/make_golden_eval.py -> reference_eval_outputs.npz)."""
import os
import sys

import numpy as np
import pytest

import fake_backend
import golden_inputs as GI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _init_paths  # noqa: F401,E402


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_eval_outputs.npz")))


@pytest.fixture()
def devkit(tmp_path):
    case = GI.sds_case()
    root = str(tmp_path / "VOCdevkitSDS")
    GI.write_sds_devkit(root, case)
    return root, case


def test_voc_eval_sds_matches_the_reference(devkit, ref, tmp_path):
    from datasets.pascal_voc_seg import PascalVOCSeg
    root, case = devkit
    imdb = PascalVOCSeg("val", "2012", root, image_ext=".npy")
    assert imdb.name == "voc_2012_val" and imdb.num_classes == 21 and len(imdb.image_index) == 6
    assert imdb.image_path_at(2).endswith("img/syn_002.npy")
    out = str(tmp_path / "out")
    os.mkdir(out)
    with np.errstate(all="ignore"):
        res = imdb.evaluate_segmentation(case["pred_boxes"], case["pred_masks"], out)
    assert np.array_equal(np.array(res[0.5]), ref["eval_ap_05"], equal_nan=True)
    assert np.array_equal(np.array(res[0.7]), ref["eval_ap_07"], equal_nan=True)
    assert 0.1 < ref["eval_ap_07"][4] < 0.9                          # the case is not trivially 0 / 1
    assert os.path.isfile(os.path.join(out, "aeroplane_det.pkl")) and os.path.isfile(os.path.join(out, "tvmonitor_seg.pkl"))


def test_parse_inst_and_mask_overlap(devkit):
    from transform.mask_transform import mask_overlap
    from utils.voc_eval import parse_inst, voc_ap
    root, case = devkit
    rec = parse_inst("syn_002", root)
    inst = case["images"][2]["inst"]
    assert len(rec) == len(np.unique(inst)) - 1
    for r in rec:
        x1, y1, x2, y2 = (int(v) for v in r["mask_bound"])
        assert r["mask"].shape == (y2 - y1 + 1, x2 - x1 + 1) and r["mask"].any(0).all() and r["mask"].any(1).all()
    a = np.ones((4, 6), bool)
    assert mask_overlap([0, 0, 5, 3], [0, 0, 5, 3], a, a) == 1.0
    assert mask_overlap([0, 0, 5, 3], [10, 10, 15, 13], a, a) == 0
    assert abs(mask_overlap([0, 0, 5, 3], [3, 0, 8, 3], a, a) - 12.0 / 36.0) < 1e-12
    assert abs(voc_ap(np.array([0.5, 1.0]), np.array([1.0, 0.5]), True) - (6 * 1.0 + 5 * 0.5) / 11) < 1e-12
    assert abs(voc_ap(np.array([0.5, 1.0]), np.array([1.0, 0.5]), False) - 0.75) < 1e-12


def test_tester_wrapper_loop_matches_the_reference(devkit, ref, monkeypatch, tmp_path):
    """get_segmentation_result with a fake net leaving canned blobs: image read, prepare args (resize factors, im_info),
    un-scale / clip / concat of both stages, mask voting (the test double of the C ABI = the oracle), result lists."""
    import gc
    fake_backend.install(monkeypatch)
    import caffe
    from caffeWrapper.TesterWrapper import TesterWrapper
    from datasets.pascal_voc_seg import PascalVOCSeg
    from mnc_config import cfg
    root, case = devkit
    canned = GI.tester_net_outputs(case)

    class Blob(object):
        def __init__(self):
            self.data = np.zeros((1,), np.float32)

        def reshape(self, *dims):
            self.data = np.zeros(dims, np.float32)

    class FakeNet(object):
        def __init__(self, *a):
            self.blobs = {k: Blob() for k in list(canned[0]) + ["data", "im_info"]}
            self.calls, self.seen, self.name = 0, [], "fake"

        def forward(self, **kw):
            self.seen.append((kw["data"].shape, kw["im_info"].copy()))
            for k, v in canned[self.calls].items():
                self.blobs[k].data = v.copy()
            self.calls += 1
            return {}

    monkeypatch.setattr(caffe, "Net", FakeNet)
    monkeypatch.setattr(cfg, "ROOT_DIR", str(tmp_path))
    imdb = PascalVOCSeg("val", "2012", root, image_ext=".npy")
    t = TesterWrapper("x.prototxt", imdb, "fake.caffemodel", "seg")
    assert t.output_dir == os.path.join(str(tmp_path), "output", "default", "voc_2012_val", "fake")
    all_boxes, all_masks = t.get_segmentation_result()
    n = len(case["images"])
    assert np.array_equal(np.concatenate([s[1] for s in t.net.seen], 0), ref["tester_im_info"])
    assert np.array_equal(np.array([s[0] for s in t.net.seen]), ref["tester_data_shapes"])
    assert np.array_equal(np.array([[len(all_boxes[c][i]) for i in range(n)] for c in range(1, 21)]), ref["tester_counts"])
    boxes = np.concatenate([all_boxes[c][i] for c in range(1, 21) for i in range(n)], 0)
    masks = np.concatenate([all_masks[c][i] for c in range(1, 21) for i in range(n)], 0)
    assert boxes.dtype == ref["tester_boxes"].dtype and np.array_equal(boxes, ref["tester_boxes"])
    assert np.array_equal(masks, ref["tester_masks"])
    # get_result: pickles + evaluation run end to end on the lists (and are re-used on a second call)
    t.net.calls = 0
    with np.errstate(all="ignore"):
        res = t.get_result()
    assert set(res) == {0.5, 0.7} and len(res[0.5]) == 20
    assert os.path.isfile(os.path.join(t.output_dir, "res_boxes.pkl")) and os.path.isfile(os.path.join(t.output_dir, "res_masks.pkl"))
    calls = t.net.calls
    with np.errstate(all="ignore"):
        t.get_result()
    assert t.net.calls == calls
    gc.collect()


def test_cfm_task_matches_the_reference(devkit, ref, monkeypatch, tmp_path):
    """`--task cfm` against the reference's get_cfm_result / cfm_network_forward run on the same synthetic MCG
    maskdb with the same deterministic stand-in net: every forward's pyramid blob, rois (level index after the reference's
    min-present-level shift, scaled boxes, chunking by MAX_ROIS_GPU) and binarised masks, then the per-class heap / top-k /
    NMS-with-masks bookkeeping."""
    import gc
    fake_backend.install(monkeypatch)
    import caffe
    from caffeWrapper.TesterWrapper import TesterWrapper
    from datasets.pascal_voc_seg import PascalVOCSeg
    from mnc_config import cfg
    root, case = devkit
    GI.write_mcg_maskdb(str(tmp_path / "maskdb"), case, GI.cfm_case(case))

    class Blob(object):
        def reshape(self, *dims):
            self.shape = dims

    class FakeCfmNet(object):
        def __init__(self, *a):
            self.blobs = {k: Blob() for k in ["data", "rois", "masks"]}
            self.name, self.seen = "fakecfm", []

        def forward(self, **kw):
            assert kw["rois"].dtype == np.float32 and kw["masks"].dtype == np.float32 and kw["data"].dtype == np.float32
            assert self.blobs["data"].shape == kw["data"].shape and self.blobs["rois"].shape == kw["rois"].shape
            self.seen.append((kw["data"].shape, float(np.asarray(kw["data"], np.float64).sum()), kw["rois"].copy(),
                              kw["masks"].reshape(kw["masks"].shape[0], -1).sum(1)))
            return GI.cfm_fake_forward(kw["data"], kw["rois"], kw["masks"])

    monkeypatch.setattr(caffe, "Net", FakeCfmNet)
    monkeypatch.setattr(cfg, "ROOT_DIR", str(tmp_path))
    for k, v in GI.CFM_CFG.items():
        monkeypatch.setitem(cfg.TEST, k, v)
    monkeypatch.setitem(cfg.TEST, "MCG_MASKDB_DIR", str(tmp_path / "maskdb"))
    imdb = PascalVOCSeg("val", "2012", root, image_ext=".npy")
    t = TesterWrapper("x.prototxt", imdb, "fakecfm.caffemodel", "cfm")
    t.max_per_set, t.max_per_image = 12, 9
    cb, cm = t.get_cfm_result()
    seen = t.net.seen
    assert np.array_equal(np.array([s[0] for s in seen]), ref["cfm_data_shapes"])
    assert np.array_equal(np.array([len(s[2]) for s in seen]), ref["cfm_roi_counts"])
    assert np.array_equal(np.concatenate([s[2] for s in seen], 0), ref["cfm_rois"])
    assert np.array_equal(np.concatenate([s[3] for s in seen], 0), ref["cfm_mask_sums"])
    assert np.array_equal(np.array([s[1] for s in seen]), ref["cfm_data_sums"])
    n = len(case["images"])
    assert np.array_equal(np.array([[len(cb[c][i]) for i in range(n)] for c in range(1, 21)]), ref["cfm_counts"])
    boxes = np.concatenate([cb[c][i] for c in range(1, 21) for i in range(n)], 0)
    masks = np.concatenate([cm[c][i] for c in range(1, 21) for i in range(n)], 0)
    assert boxes.dtype == ref["cfm_boxes"].dtype and np.array_equal(boxes, ref["cfm_boxes"])
    assert masks.dtype == ref["cfm_masks"].dtype and np.array_equal(masks, ref["cfm_masks"])
    with np.errstate(all="ignore"):
        res = t.get_result()                                   # pickles + SDS evaluation on the CFM lists
    assert set(res) == {0.5, 0.7} and os.path.isfile(os.path.join(t.output_dir, "res_masks.pkl"))
    gc.collect()


def test_visualisation_tail_matches_the_reference(devkit, ref, tmp_path, monkeypatch):
    """SURVEY 8f n4 (visualisation tail): lib/utils/vis_seg.py's pure functions against the reference's own
    (_get_voc_color_map, _convert_pred_to_image incl. clipping / overwrite order / the empty negative outline slices); then the
    `vis_seg` task and demo._visualise write their files."""
    import caffe
    import demo
    from caffeWrapper.TesterWrapper import TesterWrapper
    from datasets.pascal_voc_seg import PascalVOCSeg
    from mnc_config import cfg
    from utils import vis_seg
    import pickle
    root, case = devkit
    assert np.array_equal(vis_seg._get_voc_color_map(), ref["vis_color_map"])
    for ii in (0, 3):
        H, W = case["images"][ii]["im"].shape[:2]
        inst, cls = vis_seg._convert_pred_to_image(W, H, GI.vis_pred_dict(case, ii))
        assert inst.dtype == ref["vis_inst_%d" % ii].dtype and np.array_equal(inst, ref["vis_inst_%d" % ii])
        assert np.array_equal(cls, ref["vis_cls_%d" % ii])

    class NoNet(object):
        def __init__(self, *a):
            self.name = "fake"

    monkeypatch.setattr(caffe, "Net", NoNet)
    monkeypatch.setattr(cfg, "ROOT_DIR", str(tmp_path))
    imdb = PascalVOCSeg("val", "2012", root, image_ext=".npy")
    t = TesterWrapper("x.prototxt", imdb, "fake.caffemodel", "vis_seg")
    with open(os.path.join(t.output_dir, "res_boxes.pkl"), "wb") as f:
        pickle.dump(case["pred_boxes"], f)
    with open(os.path.join(t.output_dir, "res_masks.pkl"), "wb") as f:
        pickle.dump(case["pred_masks"], f)
    t.get_result()
    from PIL import Image
    for sub, ext in (("SegInst", ".jpg"), ("SegCls", ".jpg"), ("SegRes", ".png")):
        for rec in case["images"]:
            im = Image.open(os.path.join(t.output_dir, sub, rec["name"] + ext))
            assert im.size == (rec["im"].shape[1], rec["im"].shape[0])
    res = np.asarray(Image.open(os.path.join(t.output_dir, "SegRes", case["images"][0]["name"] + ".png")).convert("RGB"))
    assert (res.astype(int) != case["images"][0]["im"][:, :, ::-1].astype(int)).any()
    out = str(tmp_path / "demo_vis.png")
    demo._visualise(case["images"][0]["im"], GI.vis_pred_dict(case, 0), out)
    assert Image.open(out).size[0] > 100
    with pytest.raises(NotImplementedError):
        TesterWrapper("x.prototxt", imdb, "fake.caffemodel", "nothing").get_result()


def test_imdb_factory():
    from db.imdb import add_imdb, get_imdb, list_imdbs
    assert "voc_2012_seg_val" in list_imdbs()
    with pytest.raises(KeyError):
        get_imdb("voc_1999_nothing")
    add_imdb("syn", lambda: 42)
    assert get_imdb("syn") == 42
