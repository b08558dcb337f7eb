"""CPU: the Faster R-CNN end2end test graph and `tools/test_net.py --task det`: the graph runs on
the MNC path's kernels unchanged (+ a test-time-identity Dropout); the wrapper loop and PASCAL VOC detection AP are pinned
against the REFERENCE'S OWN CODE on a synthetic VOCdevkit2007 (tests/golden/make_golden_eval.py)."""
import os
import sys

import numpy as np
import pytest

import fake_backend
import golden_inputs as GI
from mnc_amd import models, synth
from oracle import net as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _init_paths  # noqa: F401,E402


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_eval_outputs.npz")))


@pytest.fixture()
def fake_gpu(monkeypatch):
    import gc
    yield fake_backend.install(monkeypatch)
    gc.collect()


def test_faster_rcnn_graph_every_blob(fake_gpu):
    from mnc_amd.engine import Net
    path = models.write_faster_rcnn_end2end_test_prototxt(width_div=8)
    w = synth.synthetic_weights(path, seed=3)
    net = Net(path, w, 1, device_id=0)
    rng = np.random.default_rng(0)
    data = rng.uniform(-120, 130, (1, 3, 96, 160)).astype(np.float32)
    im_info = np.array([[96, 160, 1.0]], np.float32)
    out = net.forward(data=data, im_info=im_info)
    assert set(out) == {"cls_prob", "bbox_pred"}
    want = onet.forward_frcnn(w, data, im_info)
    for n in ("conv5_3", "rpn_bbox_pred", "rois", "pool5", "fc6", "fc7", "cls_score", "bbox_pred", "cls_prob"):
        got = net.blobs[n].data
        assert got.shape == want[n].shape, n
        assert np.abs(got - want[n]).max() <= 1e-4 * max(np.abs(want[n]).max(), 1e-6), n
    net.close()


def test_voc_eval_matches_the_reference(ref, tmp_path):
    from datasets.pascal_voc_det import PascalVOCDet
    case = GI.voc_det_case()
    root = str(tmp_path / "VOCdevkit2007")
    GI.write_voc_devkit(root, case)
    imdb = PascalVOCDet("test", "2007", root, image_ext=".npy")
    assert imdb.name == "voc_2007_test" and len(imdb.image_index) == 6 and imdb.image_path_at(1).endswith("JPEGImages/det_001.npy")
    with np.errstate(all="ignore"):
        aps = imdb.evaluate_detections(case["dets"], str(tmp_path / "out"))
    assert np.array_equal(np.array(aps), ref["det_ap"], equal_nan=True)
    assert 0.2 < np.nanmax(ref["det_ap"]) <= 1.0 + 1e-9 and len(set(np.round(ref["det_ap"][:4], 6))) > 1    # not a trivial case
    assert not [f for f in os.listdir(os.path.join(root, "results", "VOC2007", "Main")) if f.endswith(".txt")]   # cleanup
    assert os.path.isfile(str(tmp_path / "out" / "aeroplane_pr.pkl"))


def test_tester_wrapper_det_loop_matches_the_reference(ref, monkeypatch, tmp_path):
    import caffe
    from caffeWrapper.TesterWrapper import TesterWrapper
    from datasets.pascal_voc_det import PascalVOCDet
    from mnc_config import cfg
    fake_backend.install(monkeypatch)
    case = GI.voc_det_case()
    root = str(tmp_path / "VOCdevkit2007")
    GI.write_voc_devkit(root, case)
    canned = GI.tester_det_outputs(case)

    class Blob(object):
        def __init__(self):
            self.data = np.zeros((1,), np.float32)

        def reshape(self, *dims):
            self.data = np.zeros(dims, np.float32)

    class FakeNet(object):
        def __init__(self, *a):
            self.blobs = {k: Blob() for k in ("rois", "data", "im_info")}
            self.calls, self.name = 0, "fakedet"

        def forward(self, **kw):
            out = canned[self.calls]
            self.blobs["rois"].data = out["rois"].copy()
            self.calls += 1
            return {"bbox_pred": out["bbox_pred"].copy(), "cls_prob": out["cls_prob"].copy()}

    monkeypatch.setattr(caffe, "Net", FakeNet)
    monkeypatch.setattr(cfg, "ROOT_DIR", str(tmp_path))
    imdb = PascalVOCDet("test", "2007", root, image_ext=".npy")
    captured = {}
    monkeypatch.setattr(imdb, "evaluate_detections", lambda boxes, out: captured.setdefault("boxes", boxes))
    t = TesterWrapper("x.prototxt", imdb, "fakedet.caffemodel", "det")
    t.get_result()
    nd = captured["boxes"]
    n = len(case["images"])
    assert np.array_equal(np.array([[len(nd[c][i]) for i in range(n)] for c in range(1, 21)]), ref["tester_det_counts"])
    boxes = np.concatenate([nd[c][i] for c in range(1, 21) for i in range(n) if len(nd[c][i])], 0)
    assert boxes.dtype == np.float32 and np.array_equal(boxes, ref["tester_det_boxes"])
    assert os.path.isfile(os.path.join(t.output_dir, "detections.pkl"))
