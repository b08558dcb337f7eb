"""Seeded synthetic inputs shared by tests/golden/make_golden.py (which feeds them to the REFERENCE's code) and by
the tests (which feed the same arrays to the oracle and to the HIP path).  numpy Generator streams are stable
across numpy versions for the methods used here (uniform, normal, permutation, integers, choice).
"""
import os

import numpy as np

# (n boxes, threshold, seed) -- SURVEY.md section 8(d) "unit-kernel synthetic inputs"
NMS_CASES = [(600, 0.3, 10), (6000, 0.7, 11), (1, 0.5, 12), (64, 0.5, 13), (65, 0.5, 14), (200, 0.0, 15)]
# tag -> (instances, H, W, seed): a small canvas and BASELINE's 600 instances on a 600x1000 canvas
VOTING_CASES = {"small": (120, 160, 240, 20), "full": (600, 600, 1000, 21)}


def _boxes(rng, n, W, H, lo=16, hi=400):
    x1 = rng.uniform(0, W - 64, n)
    y1 = rng.uniform(0, H - 64, n)
    x2 = np.minimum(x1 + rng.uniform(lo, hi, n), W - 1)
    y2 = np.minimum(y1 + rng.uniform(lo, hi, n), H - 1)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)


def nms_case(n, seed, W=1000, H=600):
    """dets [n,5] float32 with DISTINCT scores (argsort()[::-1] of tied scores is unspecified)."""
    rng = np.random.default_rng(seed)
    b = _boxes(rng, n, W, H)
    s = rng.permutation(np.linspace(0.01, 0.99, n)).astype(np.float32) if n > 1 else np.array([0.5], np.float32)
    return np.hstack([b, s[:, None]]).astype(np.float32)


def bbox_case(seed):
    rng = np.random.default_rng(seed)
    boxes = _boxes(rng, 64, 320, 200, 4, 120)
    boxes[0] = [-5.0, -3.0, 10.0, 8.0]          # crosses the border -> clipped
    boxes[1] = [300.0, 190.0, 330.0, 210.0]
    deltas = rng.normal(0, 0.4, (64, 12)).astype(np.float32)   # K = 3 classes -> stride-4 slicing
    return {"boxes": boxes, "deltas": deltas, "im_shape": (200.0, 320.0)}


def proposal_case(fh, fw, seed):
    """Softmax-shaped RPN outputs on an fh x fw stride-16 map: channels 0..8 bg, 9..17 fg (proposal_layer.py:75)."""
    rng = np.random.default_rng(seed)
    n = 9 * fh * fw
    fg = rng.permutation(np.linspace(0.001, 0.999, n)).astype(np.float32).reshape(1, 9, fh, fw)
    cls_prob = np.concatenate([(1.0 - fg).astype(np.float32), fg], axis=1)
    bbox_pred = rng.normal(0, 0.5, (1, 36, fh, fw)).astype(np.float32)
    H = 600 if fh == 38 else fh * 16
    W = 1000 if fw == 63 else fw * 16
    im_info = np.array([[H, W, 1.0]], np.float32)
    return {"cls_prob": cls_prob, "bbox_pred": bbox_pred, "im_info": im_info}


def _softmax_rows(x):
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def stage_bridge_case(seed, R=50, H=600, W=1000):
    rng = np.random.default_rng(seed)
    rois = np.hstack([np.zeros((R, 1), np.float32), _boxes(rng, R, W, H)]).astype(np.float32)
    bbox_pred = rng.normal(0, 0.1, (R, 84)).astype(np.float32)
    scores = _softmax_rows(rng.normal(0, 2.0, (R, 21)))
    return {"rois": rois, "bbox_pred": bbox_pred, "scores": scores, "im_info": np.array([[H, W, 1.0]], np.float32)}


def _blob_masks(rng, n, S=21):
    """Sigmoid masks with an object-like blob so that weighted sums exceed the 0.4 binarisation threshold."""
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float32)
    cx = rng.uniform(7, 13, (n, 1, 1))
    cy = rng.uniform(7, 13, (n, 1, 1))
    rad = rng.uniform(4, 9, (n, 1, 1))
    logit = 3.0 * (1.0 - ((xx - cx) ** 2 + (yy - cy) ** 2) / rad ** 2) + rng.normal(0, 0.7, (n, S, S))
    return (1.0 / (1.0 + np.exp(-logit))).astype(np.float32).reshape(n, 1, S, S)


def voting_case(n, H, W, seed, n_obj=8):
    """`n` instances jittered around `n_obj` objects: boxes [n,4] f32 (original-image pixels), masks [n,1,21,21],
    scores [n,21] (softmax rows, the object's class dominant)."""
    rng = np.random.default_rng(seed)
    ow = rng.uniform(0.15, 0.5, n_obj) * W
    oh = rng.uniform(0.15, 0.6, n_obj) * H
    ox = rng.uniform(0, 1, n_obj) * (W - ow)
    oy = rng.uniform(0, 1, n_obj) * (H - oh)
    ocls = rng.integers(1, 21, n_obj)
    k = rng.integers(0, n_obj, n)
    jit = rng.normal(0, 0.06, (n, 4)) * np.stack([ow[k], oh[k], ow[k], oh[k]], 1)
    b = np.stack([ox[k], oy[k], ox[k] + ow[k], oy[k] + oh[k]], 1) + jit
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    b[:, 2] = np.maximum(b[:, 2], b[:, 0] + 2)
    b[:, 3] = np.maximum(b[:, 3], b[:, 1] + 2)
    b[:, 2] = np.minimum(b[:, 2], W - 1)
    b[:, 3] = np.minimum(b[:, 3], H - 1)
    logits = rng.normal(0, 1.0, (n, 21))
    logits[np.arange(n), ocls[k]] += rng.uniform(1.0, 6.0, n)
    return {"boxes": b.astype(np.float32), "masks": _blob_masks(rng, n), "scores": _softmax_rows(logits)}


def mv_case(seed, H=120, W=200):
    """Direct mv() input: clustered boxes, 7 results incl. one whose masks never reach 0.4 (empty -> W/2,H/2
    defaults, mv_kernel.cu:149,173), one touching the right/bottom image border, one single-candidate result."""
    rng = np.random.default_rng(seed)
    n = 40
    base = np.array([[10, 10, 80, 70], [100, 30, 199, 119], [40, 60, 120, 110], [150, 5, 190, 40]], np.float32)
    k = np.arange(n) % 4
    b = base[k] + rng.normal(0, 2.0, (n, 4)).astype(np.float32)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    b[1] = [100.0, 30.0, 199.0, 119.0]           # exactly on the image border
    masks = _blob_masks(rng, n)
    masks[k == 3] *= 0.3                          # cluster 3 never exceeds 0.4 after weighting
    inds, start, wts = [], [], []
    groups = [np.where(k == 0)[0], np.where(k == 1)[0], np.where(k == 2)[0], np.where(k == 3)[0],
              np.array([1]), np.where(k == 1)[0][:3], np.concatenate([np.where(k == 0)[0][:4], np.where(k == 2)[0][:4]])]
    for gidx in groups:
        w = rng.uniform(0.2, 1.0, len(gidx)).astype(np.float32)
        w = w / w.sum()
        inds.extend(gidx)
        wts.extend(w)
        start.append(len(inds))
    return {"boxes": b.astype(np.float32), "masks": masks, "inds": np.array(inds, np.int32),
            "start": np.array(start, np.int32), "weights": np.array(wts, np.float32), "H": H, "W": W}


def detect_case(seed, R=40, H=600, W=1000):
    """A 600x1000 uint8 BGR image (scale factor exactly 1.0) and the six blobs im_detect reads (demo.py:84-90)."""
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)

    def stage():
        rois = np.hstack([np.zeros((R, 1), np.float32), _boxes(rng, R, W + 40, H + 40)]).astype(np.float32)
        return rois, _blob_masks(rng, R), _softmax_rows(rng.normal(0, 2.0, (R, 21)))

    r1, m1, s1 = stage()
    r2, m2, s2 = stage()
    return {"im": im, "blobs": {"rois": r1, "mask_proposal": m1, "seg_cls_prob": s1,
                                "rois_ext": r2, "mask_proposal_ext": m2, "seg_cls_prob_ext": s2}}


# ---- a synthetic VOCdevkitSDS (test_net.py --task seg + voc_eval_sds) --------------------------------
SDS_IMAGES = [("syn_%03d" % i, 90 + 10 * (i % 3), 120 + 16 * (i % 4)) for i in range(6)]     # (name, H, W)


def sds_case(seed=11):
    """Per image: uint8 BGR pixels, an instance-id map and a class-id map (SBD's inst/ and cls/ .mat contents) with 1-3
    rectangular / elliptic instances, plus scored predictions per class derived from the ground truth (jittered true
    positives at several qualities, duplicates, and false positives)."""
    rng = np.random.default_rng(seed)
    case = {"images": [], "pred_boxes": None, "pred_masks": None}
    nimg = len(SDS_IMAGES)
    boxes = [[np.zeros((0, 5), np.float32) for _ in range(nimg)] for _ in range(21)]
    masks = [[np.zeros((0, 1, 21, 21), np.float32) for _ in range(nimg)] for _ in range(21)]
    yy, xx = np.mgrid[0:21, 0:21]
    for ii, (name, H, W) in enumerate(SDS_IMAGES):
        inst = np.zeros((H, W), np.uint8)
        cls = np.zeros((H, W), np.uint8)
        for k in range(1 + ii % 3):
            w, h = int(rng.integers(20, 50)), int(rng.integers(20, 45))
            x1, y1 = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
            c = int(rng.integers(1, 6))                                 # classes 1..5 only: most classes have no GT
            Y, X = np.mgrid[0:h, 0:w]
            shape = np.ones((h, w), bool) if k % 2 == 0 else \
                (((X - (w - 1) / 2.0) / (w / 2.0)) ** 2 + ((Y - (h - 1) / 2.0) / (h / 2.0)) ** 2 <= 1.0)
            region = inst[y1:y1 + h, x1:x1 + w]
            free = shape & (region == 0)
            if free.sum() < 50:
                continue
            region[free] = k + 1
            cls[y1:y1 + h, x1:x1 + w][free] = c
            # predictions for this instance: a good one, a sloppy one (shifted), and a duplicate of the good one
            for q, (dx, dy, sc) in enumerate([(1, -1, 0.9), (9, 7, 0.6), (0, 1, 0.5)]):
                bx = np.array([x1 + dx, y1 + dy, x1 + w - 1 + dx, y1 + h - 1 + dy], np.float32)
                bx[0::2] = np.clip(bx[0::2], 0, W - 1)
                bx[1::2] = np.clip(bx[1::2], 0, H - 1)
                score = np.float32(sc * rng.uniform(0.8, 1.0))
                soft = np.full((21, 21), 0.8, np.float32) if k % 2 == 0 else \
                    (0.9 - 0.9 * np.hypot((xx - 10) / 11.0, (yy - 10) / 11.0)).astype(np.float32).clip(0, 1) + 0.2
                soft = (soft + rng.normal(0, 0.03, (21, 21))).astype(np.float32)
                boxes[c][ii] = np.vstack([boxes[c][ii], np.append(bx, score)[None].astype(np.float32)])
                masks[c][ii] = np.concatenate([masks[c][ii], soft[None, None]], 0)
        for _ in range(2):                                              # false positives, one in a GT-free class
            c = int(rng.integers(1, 9))
            bx = _boxes(rng, 1, W, H, lo=10, hi=40)[0]
            boxes[c][ii] = np.vstack([boxes[c][ii], np.append(bx, np.float32(rng.uniform(0.1, 0.95)))[None].astype(np.float32)])
            masks[c][ii] = np.concatenate([masks[c][ii], rng.uniform(0, 1, (1, 1, 21, 21)).astype(np.float32)], 0)
        case["images"].append({"name": name, "im": rng.integers(0, 256, (H, W, 3), dtype=np.uint8), "inst": inst, "cls": cls})
    case["pred_boxes"], case["pred_masks"] = boxes, masks
    return case


def write_sds_devkit(root, case, image_set="val"):
    """img/<name>.npy, inst/<name>.mat, cls/<name>.mat, <image_set>.txt under `root` (SBD's struct layout)."""
    import scipy.io as sio
    for d in ("img", "inst", "cls"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for rec in case["images"]:
        np.save(os.path.join(root, "img", rec["name"] + ".npy"), rec["im"])
        sio.savemat(os.path.join(root, "inst", rec["name"] + ".mat"), {"GTinst": {"Segmentation": rec["inst"]}})
        sio.savemat(os.path.join(root, "cls", rec["name"] + ".mat"), {"GTcls": {"Segmentation": rec["cls"]}})
    with open(os.path.join(root, image_set + ".txt"), "w") as f:
        f.write("".join(rec["name"] + "\n" for rec in case["images"]))


def tester_net_outputs(case, seed=12, R=30):
    """What a net would leave in its blobs for each image of the case (two stages of R rois in the RESIZED image's
    coordinates, masks, class probabilities) -- the canned outputs of the fake net that drives TesterWrapper."""
    outs = []
    for ii, rec in enumerate(case["images"]):
        H, W = rec["im"].shape[:2]
        scale = min(600.0 / min(H, W), 1000.0 / max(H, W))
        vc = voting_case(2 * R, H, W, seed + ii, n_obj=4)
        rois = np.hstack([np.zeros((2 * R, 1), np.float32), vc["boxes"] * np.float32(scale)]).astype(np.float32)
        outs.append({"rois": rois[:R], "mask_proposal": vc["masks"][:R], "seg_cls_prob": vc["scores"][:R],
                     "rois_ext": rois[R:], "mask_proposal_ext": vc["masks"][R:], "seg_cls_prob_ext": vc["scores"][R:]})
    return outs


# ---- a synthetic VOCdevkit2007 (test_net.py --task det + voc_eval) -----------------------------------------
VOC_CLASSES = ('__background__', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


def voc_det_case(seed=21):
    """Per image: uint8 BGR pixels and annotated objects (class, difficult flag, 1-based inclusive bbox); plus per-class
    scored detections [n,5] (0-based x1,y1,x2,y2,score): jittered true positives, duplicates, false positives, and hits on
    'difficult' objects."""
    rng = np.random.default_rng(seed)
    nimg = len(SDS_IMAGES)
    dets = [[np.zeros((0, 5), np.float32) for _ in range(nimg)] for _ in range(21)]
    images = []
    for ii, (name, H, W) in enumerate(SDS_IMAGES):
        objs = []
        for k in range(1 + (ii + 1) % 3):
            w, h = int(rng.integers(20, 60)), int(rng.integers(20, 50))
            x1, y1 = int(rng.integers(1, W - w)), int(rng.integers(1, H - h))
            c = int(rng.integers(1, 5))
            difficult = int(k == 2)
            objs.append({"name": VOC_CLASSES[c], "difficult": difficult, "bbox": [x1, y1, x1 + w, y1 + h]})
            for dx, dy, sc in [(1, 0, 0.95), (8, 9, 0.55), (-1, 1, 0.5)]:
                b = np.array([x1 - 1 + dx, y1 - 1 + dy, x1 + w - 1 + dx, y1 + h - 1 + dy, sc * rng.uniform(0.8, 1.0)], np.float32)
                dets[c][ii] = np.vstack([dets[c][ii], b[None]])
        for _ in range(2):
            c = int(rng.integers(1, 8))
            b = np.append(_boxes(rng, 1, W, H, lo=10, hi=40)[0], np.float32(rng.uniform(0.1, 0.9))).astype(np.float32)
            dets[c][ii] = np.vstack([dets[c][ii], b[None]])
        images.append({"name": name.replace("syn", "det"), "im": rng.integers(0, 256, (H, W, 3), dtype=np.uint8), "objects": objs,
                       "H": H, "W": W})
    for c in range(1, 21):          # every class reports something (the reference's voc_eval cannot read an empty file)
        b = np.append(_boxes(rng, 1, SDS_IMAGES[0][2], SDS_IMAGES[0][1], lo=10, hi=30)[0], np.float32(0.05)).astype(np.float32)
        dets[c][0] = np.vstack([dets[c][0], b[None]])
    return {"images": images, "dets": dets}


def write_voc_devkit(root, case, year="2007", image_set="test"):
    """VOC<year>/JPEGImages/<name>.npy, VOC<year>/Annotations/<name>.xml, VOC<year>/ImageSets/Main/<set>.txt under `root`."""
    base = os.path.join(root, "VOC" + year)
    for d in ("JPEGImages", "Annotations", os.path.join("ImageSets", "Main")):
        os.makedirs(os.path.join(base, d), exist_ok=True)
    for rec in case["images"]:
        np.save(os.path.join(base, "JPEGImages", rec["name"] + ".npy"), rec["im"])
        objs = "".join(
            "<object><name>%s</name><pose>Unspecified</pose><truncated>0</truncated><difficult>%d</difficult>"
            "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
            % ((o["name"], o["difficult"]) + tuple(o["bbox"])) for o in rec["objects"])
        with open(os.path.join(base, "Annotations", rec["name"] + ".xml"), "w") as f:
            f.write("<annotation><filename>%s.jpg</filename><size><width>%d</width><height>%d</height><depth>3</depth></size>%s"
                    "</annotation>" % (rec["name"], rec["W"], rec["H"], objs))
    with open(os.path.join(base, "ImageSets", "Main", image_set + ".txt"), "w") as f:
        f.write("".join(rec["name"] + "\n" for rec in case["images"]))


def tester_det_outputs(case, seed=22, R=40):
    """Canned blobs of a Faster R-CNN net per image: rois [R,5] (resized-image coordinates), bbox_pred [R,84], cls_prob [R,21]."""
    outs = []
    for ii, rec in enumerate(case["images"]):
        rng = np.random.default_rng(seed + ii)
        H, W = rec["H"], rec["W"]
        scale = min(600.0 / min(H, W), 1000.0 / max(H, W))
        rois = np.hstack([np.zeros((R, 1), np.float32), _boxes(rng, R, W, H, lo=12, hi=60) * np.float32(scale)]).astype(np.float32)
        outs.append({"rois": rois, "bbox_pred": rng.normal(0, 0.2, (R, 84)).astype(np.float32),
                     "cls_prob": _softmax_rows(rng.normal(0, 2.0, (R, 21)))})
    return outs


# ---- CFM test path (test_net.py --task cfm): MCG-style proposals + a deterministic stand-in for the net ----
CFM_CFG = {"SCALES": [120, 160, 220, 320, 440], "MAX_SIZE": 640, "GROUP_SCALE": 3, "MAX_ROIS_GPU": [7, 5], "USE_TOP_K_MCG": 30}


def cfm_case(case, seed=31):
    """Per image of an sds_case: {'boxes': [n,4] float64, 'masks': [n,21,21] bool} as tools/prepare_mcg_maskdb.py:88-94 writes
    them for the val set -- sizes from below the 16-pixel filter up to most of the image, so that the proposals spread over
    all pyramid levels; more rows than USE_TOP_K_MCG keeps."""
    rng = np.random.default_rng(seed)
    out = []
    yy, xx = np.mgrid[0:21, 0:21]
    for ii, rec in enumerate(case["images"]):
        H, W = rec["im"].shape[:2]
        n = 34 + ii
        w = rng.integers(10, W - 4, n)
        h = rng.integers(10, H - 4, n)
        w[:4], h[:4] = [12, 40, 15, 16], [40, 12, 15, 16]        # three rows under the min_size=16 filter, one exactly on it
        x1 = (rng.uniform(0, 1, n) * (W - w)).astype(np.int64)
        y1 = (rng.uniform(0, 1, n) * (H - h)).astype(np.int64)
        boxes = np.stack([x1, y1, x1 + w - 1, y1 + h - 1], 1).astype(np.float64)
        cx, cy, r = rng.uniform(5, 15, n), rng.uniform(5, 15, n), rng.uniform(4, 12, n)
        masks = ((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) <= r[:, None, None] ** 2
        out.append({"boxes": boxes, "masks": masks})
    return out


def write_mcg_maskdb(directory, case, mcg):
    import scipy.io
    os.makedirs(directory, exist_ok=True)
    for rec, db in zip(case["images"], mcg):
        scipy.io.savemat(os.path.join(directory, rec["name"] + ".mat"), {"boxes": db["boxes"], "masks": db["masks"]})


def cfm_fake_forward(data, rois, masks, K=21, S=21):
    """Deterministic stand-in for the CFM net: outputs are smooth functions of the INPUTS the tester built (level index and
    scaled box per roi, the binarised mask, the pyramid blob), so any difference in those inputs changes the results."""
    rois = np.asarray(rois, np.float64)
    m = np.asarray(masks, np.float64).reshape(rois.shape[0], -1)
    dsum = float(np.asarray(data, np.float64).mean())
    key = rois[:, 1] * 0.013 + rois[:, 2] * 0.017 + rois[:, 3] * 0.007 + rois[:, 4] * 0.011 + rois[:, 0] * 0.5 + m.sum(1) * 0.03 + dsum
    logits = 3.0 * np.sin(key[:, None] * (1.0 + 0.37 * np.arange(K)[None, :]))
    e = np.exp(logits - logits.max(1, keepdims=True))
    seg = (e / e.sum(1, keepdims=True)).astype(np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    mp = 0.5 + 0.5 * np.sin(key[:, None, None] + 0.3 * xx[None] + 0.2 * yy[None] + m.mean(1)[:, None, None])
    return {"mask_prob": mp.reshape(-1, 1, S, S).astype(np.float32), "seg_cls_prob": seg}


def vis_pred_dict(case, ii, vis_thresh=0.3):
    """What lib/utils/vis_seg.py:_prepare_dict builds for image ii from the case's predictions (plus one box touching the
    image border and one hanging over it, to reach the clipping and the negative-slice outline branches)."""
    boxes, masks, classes = [], [], []
    for c in range(1, 21):
        det, seg = case["pred_boxes"][c][ii], case["pred_masks"][c][ii]
        for k in np.where(det[:, -1] >= vis_thresh)[0]:
            boxes.append(det[k])
            masks.append(seg[k][0])
            classes.append(c)
    H, W = case["images"][ii]["im"].shape[:2]
    yy, xx = np.mgrid[0:21, 0:21]
    disc = (np.hypot(xx - 10, yy - 10) <= 8).astype(np.float32)
    boxes += [np.array([0, 0, 30.4, 25.6, 0.9], np.float32), np.array([W - 20.5, H - 18.5, W + 9, H + 7, 0.8], np.float32)]
    masks += [disc, disc * np.float32(0.7)]
    classes += [7, 15]
    return {"image_name": case["images"][ii]["name"], "cls_name": classes, "boxes": boxes, "masks": masks}
