"""GPU: the reference demo's real-image plumbing (tools/demo.py:131-149 of the reference: two warm-ups, then every JPEG through
cv2.imread -> prepare_mnc_args (resize by 600 / short side) -> net.forward -> un-scale / clip -> gpu_mask_voting -> visualisation)
on real JPEG photographs.  The reference's five data/demo/*.jpg cannot travel to the GPU box (nothing under /root/reference exists
there), so the test takes the natural-image JPEGs that ship inside this image's Python packages (scikit-learn's china.jpg and
flower.jpg, 427x640 -> scale 1.405; matplotlib's grace_hopper.jpg, 600x512 -> scale 1.172) -- and the reference's own five as
well wherever /root/reference is mounted.

Checked per image: the PIL-decoded BGR image prepared on the GPU == the oracle's prep_im_for_blob (cv2.resize restatement,
oracle/SPEC.md section 5) bit for bit; boxes == oracle im_detect tail on the device's rois (un-scaling by the non-unit scale,
clipping to the ORIGINAL image); voted instances == oracle gpu_mask_voting on the device's outputs, bit-exact; the PNG is
written.  Trained weights are not available here: detections are those of seeded random weights (plumbing, not accuracy)."""
import glob
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

import mnc_amd
from oracle import host as ohost
from test_gpu_engine import _log

pytestmark = pytest.mark.gpu
mnc_amd.install_paths()


def _jpegs():
    out = []
    try:
        import sklearn
        d = os.path.join(os.path.dirname(sklearn.__file__), "datasets", "images")
        out += [os.path.join(d, f) for f in ("china.jpg", "flower.jpg")]
    except ImportError:
        pass
    try:
        import matplotlib
        out.append(os.path.join(matplotlib.get_data_path(), "sample_data", "grace_hopper.jpg"))
    except ImportError:
        pass
    out += sorted(glob.glob("/root/reference/data/demo/*.jpg"))
    return [p for p in out if os.path.isfile(p)]


def test_demo_on_real_jpegs(tmp_path):
    import demo
    from mnc_amd import models, synth
    from transform.mask_transform import gpu_mask_voting
    images = _jpegs()
    if not images:
        pytest.skip("no JPEG photographs on this box")
    # (1) the entry point as a user runs it: flags, warm-ups, per-image loop, PNGs
    buf = io.StringIO()
    with redirect_stdout(buf):
        demo.main(["--images"] + images + ["--out-dir", str(tmp_path), "--vis-thresh", "0.0"])
    text = buf.getvalue()
    assert text.count("Demo for") == len(images) and text.count("forward time") == len(images)
    pngs = sorted(os.listdir(str(tmp_path)))
    assert len(pngs) == len(images) and all(os.path.getsize(os.path.join(str(tmp_path), p)) > 10000 for p in pngs)
    lines = [l for l in text.splitlines() if l.startswith(("Demo for", "forward time", "mask voting time")) or "instances with" in l]
    # (2) the same per-image body against the oracle
    import caffe
    proto = models.write_mnc_5stage_test_prototxt()
    w = synth.synthetic_weights(proto, seed=0)
    net = caffe.Net(proto, w, caffe.TEST)
    try:
        for path in images:
            im = demo._read_image_bgr(path)
            assert im.dtype == np.uint8 and im.ndim == 3
            data, im_info, scale = ohost.prepare_mnc_args(im)
            assert scale != 1.0
            boxes, masks, scores = demo.im_detect(im, net)
            got = net.blobs["data"]._host_read()
            assert got.shape == data.shape and np.array_equal(got, data), path            # decode -> resize -> mean -> layout
            g = lambda n: net.blobs[n]._host_read()
            ob, om, osc = ohost.im_detect_tail(g("rois"), g("mask_proposal"), g("seg_cls_prob"), g("rois_ext"), g("mask_proposal_ext"),
                                               g("seg_cls_prob_ext"), scale, im.shape)
            assert np.array_equal(np.asarray(boxes), ob) and np.array_equal(np.asarray(masks), om) and np.array_equal(np.asarray(scores), osc)
            lm, lb = gpu_mask_voting(masks, boxes, scores, 21, 100, im.shape[1], im.shape[0])
            wm, wb = ohost.gpu_mask_voting(om, ob, osc, 21, 100, im.shape[1], im.shape[0])
            assert np.array_equal(np.concatenate(lb, 0), np.concatenate(wb, 0))
            assert np.array_equal(np.concatenate(lm, 0), np.concatenate(wm, 0), equal_nan=True)
            lines.append("%s: %dx%d -> net input %dx%d (scale %.4f), %d voted instances == oracle voting; prep == oracle prep bit for bit"
                         % (os.path.basename(path), im.shape[0], im.shape[1], data.shape[2], data.shape[3], scale, sum(len(b) for b in lb)))
    finally:
        net.close()
    print("\n".join(lines))
    _log(lines)
