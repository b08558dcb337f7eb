#!/usr/bin/env python3
"""Golden outputs for the caller on the output side of the hot path, produced by RUNNING THE REFERENCE'S OWN
CODE in this container (same in-memory python-2 -> 3 patching and stubs as make_golden.py):

  * lib/utils/voc_eval.py:voc_eval_sds (with parse_inst / check_voc_sds_cache / voc_ap and
    lib/transform/mask_transform.py:mask_overlap) on the synthetic VOCdevkitSDS of tests/golden_inputs.py:sds_case, with
    the result pickles written the way lib/datasets/pascal_voc_seg.py:_write_voc_seg_results_file does
    -> eval_ap_05 / eval_ap_07 [20];
  * lib/caffeWrapper/TesterWrapper.py:get_segmentation_result (and _segmentation_forward / _prepare_mnc_args) driven by a
    fake caffe.Net that leaves canned blobs (golden_inputs.tester_net_outputs), with gpu_mask_voting = the reference's
    mask_transform.gpu_mask_voting over oracle/_ref  -> tester_* arrays.

The reference targets numpy 1.x: where its result dtype depends on value-based casting (`rois / im_scales[0]`) the legacy
rule is applied explicitly.  cv2 is absent: cv2.resize is the oracle's restatement (oracle/host.py), cv2.imread reads the .npy images.

    python tests/golden/make_golden_eval.py        -> tests/golden/reference_eval_outputs.npz
"""
import os
import pickle
import re
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import golden_inputs as GI  # noqa: E402
from oracle import host as ohost  # noqa: E402


def _join_multiline_prints(src):
    """`print 'x'.format(\n   a, b)` -> one physical line, so that make_golden._py3's print rule applies."""
    out, lines, i = [], src.split("\n"), 0
    while i < len(lines):
        line = lines[i]
        if re.match(r"^\s*print ", line):
            while line.count("(") > line.count(")") and i + 1 < len(lines):
                i += 1
                line = line.rstrip() + " " + lines[i].strip()
        out.append(line)
        i += 1
    return "\n".join(out)


def _py3_more(src):
    src = MG._py3(_join_multiline_prints(src))
    src = src.replace("import cPickle", "import pickle as cPickle")
    src = re.sub(r"open\((\w+), 'wr?'\) as f:(\s+)cPickle\.dump", r"open(\1, 'wb') as f:\2cPickle.dump", src)
    src = re.sub(r"open\((\w+), 'r'\) as f:(\s+)(\w+) = cPickle\.load", r"open(\1, 'rb') as f:\2\3 = cPickle.load", src)
    src = src.replace("astype(np.bool)", "astype(bool)")
    # numpy-1.x value-based casting (what the reference ran under): float32 array / 0-d float64 array stays float32; numpy 2
    # would promote to float64
    src = src.replace("/ im_scales[0]", "/ np.float32(im_scales[0])")
    src = src.replace("mask_bound[1]:mask_bound[3]+1, mask_bound[0]:mask_bound[2]+1",
                      "int(mask_bound[1]):int(mask_bound[3])+1, int(mask_bound[0]):int(mask_bound[2])+1")   # float slice indices
    return src


def _load(name, relpath):
    path = os.path.join(MG.REF, relpath)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    with open(path) as f:
        exec(compile(_py3_more(f.read()), path, "exec"), mod.__dict__)
    return mod


def main():
    R = MG.install_reference()

    def resize(im, dsize, dst=None, fx=None, fy=None, interpolation=None):
        if dsize is None:
            return ohost.resize_bilinear_cv(im, fx, fy)
        return ohost.resize_bilinear_cv_to(im, dsize[0], dsize[1])

    sys.modules["cv2"].resize = resize
    sys.modules["cv2"].imread = lambda path: np.load(path)
    for m in (R.blob, R.demo):
        if hasattr(m, "cv2"):
            m.cv2.resize = resize
    g = {}
    case = GI.sds_case()

    # ---- 1. voc_eval_sds ------------------------------------------------------------------------------------------
    voc_eval = _load("utils.voc_eval", "lib/utils/voc_eval.py")
    classes = ('__background__', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')
    with tempfile.TemporaryDirectory() as root:
        GI.write_sds_devkit(root, case)
        out = os.path.join(root, "out")
        os.mkdir(out)
        # pascal_voc_seg.py:_reformat_result + _write_voc_seg_results_file
        for c, cname in enumerate(classes):
            if c == 0:
                continue
            with open(os.path.join(out, cname + "_det.pkl"), "wb") as f:
                pickle.dump(case["pred_boxes"][c], f)
            seg = [m.reshape(m.shape[0], 21, 21) >= R.cfg.BINARIZE_THRESH if len(m) else [] for m in case["pred_masks"][c]]
            with open(os.path.join(out, cname + "_seg.pkl"), "wb") as f:
                pickle.dump(seg, f)
        for thr, key in ((0.5, "eval_ap_05"), (0.7, "eval_ap_07")):
            cache = os.path.join(root, "cache_%s" % key)
            aps = []
            for cname in classes[1:]:
                with np.errstate(all="ignore"):
                    aps.append(voc_eval.voc_eval_sds(os.path.join(out, cname + "_det.pkl"), os.path.join(out, cname + "_seg.pkl"),
                                                     root, os.path.join(root, "val.txt"), cname, cache, classes, ov_thresh=thr))
            g[key] = np.array(aps, np.float64)

        # ---- 2. TesterWrapper.get_segmentation_result with a fake net --------------------------------------------------
        canned = GI.tester_net_outputs(case)

        class FakeNet(object):
            def __init__(self, *a):
                self.blobs = {k: MG.Blob() for k in list(canned[0]) + ["data", "im_info"]}
                self.calls = 0
                self.name = "fake"
                self.seen = []

            def forward(self, **kw):
                self.seen.append((kw["data"].shape, kw["im_info"].copy()))
                for k, v in canned[self.calls].items():
                    self.blobs[k].data = v.copy()
                self.calls += 1
                return {}

        sys.modules["caffe"].Net = FakeNet
        _load("utils.timer", "lib/utils/timer.py")
        tw = _load("caffeWrapper.TesterWrapper", "lib/caffeWrapper/TesterWrapper.py")

        names = [r["name"] for r in case["images"]]

        class Imdb(object):
            name = "syn_sds_val"
            image_index = names
            num_classes = 21

            def image_path_at(self, i):
                return os.path.join(root, "img", self.image_index[i] + ".npy")

        R.cfg.ROOT_DIR = root
        t = tw.TesterWrapper("x.prototxt", Imdb(), "fake.caffemodel", "seg")
        all_boxes, all_masks = t.get_segmentation_result()
        g["tester_im_info"] = np.concatenate([s[1] for s in t.net.seen], 0)
        g["tester_data_shapes"] = np.array([s[0] for s in t.net.seen], np.int64)
        g["tester_counts"] = np.array([[len(all_boxes[c][i]) for i in range(len(case["images"]))] for c in range(1, 21)])
        g["tester_boxes"] = np.concatenate([all_boxes[c][i] for c in range(1, 21) for i in range(len(case["images"]))], 0)
        g["tester_masks"] = np.concatenate([all_masks[c][i] for c in range(1, 21) for i in range(len(case["images"]))], 0)
        # ---- 2b. CFM task: TesterWrapper.get_cfm_result / cfm_network_forward with a deterministic net --------
        mcg = GI.cfm_case(case)
        cwd = os.getcwd()
        os.chdir(root)                                   # the reference reads 'data/cache/voc_2012_val_mcg_maskdb/' from the cwd
        try:
            GI.write_mcg_maskdb(os.path.join(root, "data", "cache", "voc_2012_val_mcg_maskdb"), case, mcg)

            class FakeCfmNet(object):
                def __init__(self, *a):
                    self.blobs = {k: MG.Blob() for k in ["data", "rois", "masks"]}
                    self.name, self.seen = "fakecfm", []

                def forward(self, **kw):
                    self.seen.append((kw["data"].shape, float(np.asarray(kw["data"], np.float64).sum()), kw["rois"].copy(),
                                      kw["masks"].reshape(kw["masks"].shape[0], -1).sum(1)))
                    assert kw["rois"].dtype == np.float32 and kw["masks"].dtype == np.float32
                    return GI.cfm_fake_forward(kw["data"], kw["rois"], kw["masks"])

            sys.modules["caffe"].Net = FakeCfmNet
            saved = {k: R.cfg.TEST[k] for k in GI.CFM_CFG}
            for k, v in GI.CFM_CFG.items():
                R.cfg.TEST[k] = v
            tw.cv2.resize = resize
            tw.cv2.imread = lambda path: np.load(path)

            class CfmImdb(Imdb):
                _image_index = names

            t = tw.TesterWrapper("x.prototxt", CfmImdb(), "fakecfm.caffemodel", "cfm")
            t.max_per_set = 12                                   # small enough for the per-class score heap to bite
            t.max_per_image = 9
            cb, cm = t.get_cfm_result()
            g["cfm_data_shapes"] = np.array([s[0] for s in t.net.seen], np.int64)
            g["cfm_data_sums"] = np.array([s[1] for s in t.net.seen], np.float64)
            g["cfm_rois"] = np.concatenate([s[2] for s in t.net.seen], 0)
            g["cfm_roi_counts"] = np.array([len(s[2]) for s in t.net.seen], np.int64)
            g["cfm_mask_sums"] = np.concatenate([s[3] for s in t.net.seen], 0)
            nimg = len(case["images"])
            g["cfm_counts"] = np.array([[len(cb[c][i]) for i in range(nimg)] for c in range(1, 21)])
            g["cfm_boxes"] = np.concatenate([cb[c][i] for c in range(1, 21) for i in range(nimg)], 0)
            g["cfm_masks"] = np.concatenate([cm[c][i] for c in range(1, 21) for i in range(nimg)], 0)
            for k, v in saved.items():
                R.cfg.TEST[k] = v
        finally:
            os.chdir(cwd)
    # ---- 2c. visualisation tail: lib/utils/vis_seg.py pure functions ----------------------------------------
    vs = _load("utils.vis_seg", "lib/utils/vis_seg.py")
    vs.cv2.resize = resize
    g["vis_color_map"] = vs._get_voc_color_map()
    for ii in (0, 3):
        H, W = case["images"][ii]["im"].shape[:2]
        pred = GI.vis_pred_dict(case, ii)
        inst_img, cls_img = vs._convert_pred_to_image(W, H, pred)
        g["vis_inst_%d" % ii], g["vis_cls_%d" % ii] = inst_img, cls_img
    # ---- 3. detection task: voc_eval + TesterWrapper.get_detection_result ----------------------------------
    dcase = GI.voc_det_case()
    with tempfile.TemporaryDirectory() as root:
        GI.write_voc_devkit(root, dcase)
        res = os.path.join(root, "results")
        os.mkdir(res)
        names = [r["name"] for r in dcase["images"]]
        for c, cname in enumerate(GI.VOC_CLASSES):
            if c == 0:
                continue
            with open(os.path.join(res, "det_%s.txt" % cname), "wt") as f:      # pascal_voc_det.py:_write_voc_results_file
                for im_ind, index in enumerate(names):
                    dets = dcase["dets"][c][im_ind]
                    for k in range(dets.shape[0]):
                        f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(index, dets[k, -1], dets[k, 0] + 1,
                                                                                  dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))
        aps = []
        for cname in GI.VOC_CLASSES[1:]:
            with np.errstate(all="ignore"):
                rec, prec, ap = voc_eval.voc_eval(os.path.join(res, "det_{:s}.txt"),
                                                  os.path.join(root, "VOC2007", "Annotations", "{:s}.xml"),
                                                  os.path.join(root, "VOC2007", "ImageSets", "Main", "test.txt"), cname,
                                                  os.path.join(root, "cache"), ovthresh=0.5, use_07_metric=True)
            aps.append(ap)
        g["det_ap"] = np.array(aps, np.float64)

        dcanned = GI.tester_det_outputs(dcase)

        class FakeDetNet(object):
            def __init__(self, *a):
                self.blobs = {k: MG.Blob() for k in ["rois", "data", "im_info"]}
                self.calls, self.name = 0, "fakedet"

            def forward(self, **kw):
                out = dcanned[self.calls]
                self.blobs["rois"].data = out["rois"].copy()
                self.calls += 1
                return {"bbox_pred": out["bbox_pred"].copy(), "cls_prob": out["cls_prob"].copy()}

        sys.modules["caffe"].Net = FakeDetNet
        captured = {}

        class DetImdb(object):
            name = "syn_voc_test"
            image_index = names
            num_classes = 21

            def image_path_at(self, i):
                return os.path.join(root, "VOC2007", "JPEGImages", self.image_index[i] + ".npy")

            def evaluate_detections(self, all_boxes, output_dir):
                captured["boxes"] = all_boxes

        R.cfg.ROOT_DIR = root
        t = tw.TesterWrapper("x.prototxt", DetImdb(), "fakedet.caffemodel", "det")
        t.get_result()
        nd = captured["boxes"]
        g["tester_det_counts"] = np.array([[len(nd[c][i]) for i in range(len(names))] for c in range(1, 21)])
        g["tester_det_boxes"] = np.concatenate([nd[c][i] for c in range(1, 21) for i in range(len(names)) if len(nd[c][i])], 0)
    np.savez_compressed(os.path.join(HERE, "reference_eval_outputs.npz"), **g)
    for k, v in g.items():
        print(k, v.shape, v.dtype, (np.round(v[:6] * 100, 2) if k.startswith("eval") else ""))


if __name__ == "__main__":
    main()
