#!/usr/bin/env python3
"""MNC demo on MI355X -- same flags and flow as the reference's tools/demo.py:
build the net, warm up twice on a grey image, then per image: im_detect (timed "forward time"), gpu_mask_voting,
optional visualisation.

    python tools/demo.py [--gpu 0] [--def test.prototxt] [--net weights.npz] [--images a.jpg b.jpg ...] [--no-vis]

Differences that are deliberate: weights come from an .npz (h5py is optional); without --net seeded synthetic weights
are used (the trained model cannot be fetched here), and --def defaults to the graph emitted by mnc_amd.models.
`--cpu` is accepted and ignored, exactly as in the reference (demo.py:40-42 vs :126)."""
import argparse
import os
import time

import numpy as np

import _init_paths  # noqa: F401
import caffe
from mnc_config import cfg
from transform.bbox_transform import clip_boxes
from transform.mask_transform import gpu_mask_voting
from utils.blob import can_prep_on_device, im_list_to_blob, prep_im_for_blob, prep_im_for_blob_device

CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
           "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MNC demo (MI355X)")
    p.add_argument("--gpu", dest="gpu_id", default=0, type=int, help="GPU device id to use [0]")
    p.add_argument("--cpu", dest="cpu_mode", action="store_true", help="accepted for compatibility; ignored")
    p.add_argument("--def", dest="prototxt", default=None, type=str, help="prototxt defining the network")
    p.add_argument("--net", dest="caffemodel", default=None, type=str, help="weights (.npz; .h5 needs h5py)")
    p.add_argument("--images", nargs="*", default=None, help="image files (default: data/demo/*.jpg if present)")
    p.add_argument("--no-vis", dest="vis", action="store_false", help="skip writing visualisations")
    p.add_argument("--out-dir", dest="out_dir", default=None, help="where the *_mnc.png visualisations go [next to each image]")
    p.add_argument("--vis-thresh", dest="vis_thresh", default=0.5, type=float,
                   help="score threshold of the drawn instances [0.5, the reference's constant, demo.py:103]")
    return p.parse_args(argv)


def prepare_mnc_args(im, net):
    """image (H,W,3 BGR) -> ({'data','im_info'} float32 blobs, [scale]); reshapes the two input blobs.  A uint8 image is
    mean-subtracted and resized on the GPU (cfg.TEST.DEVICE_PREP, default): `data` is then a DeviceArray holding the same
    values, which net.forward adopts without a host round trip."""
    if can_prep_on_device(net, im):
        data, im_scale = prep_im_for_blob_device(net, im, cfg.PIXEL_MEANS, cfg.TEST.SCALES[0], cfg.TRAIN.MAX_SIZE)
    else:
        im_scaled, im_scale = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TEST.SCALES[0], cfg.TRAIN.MAX_SIZE)
        data = im_list_to_blob([im_scaled]).astype(np.float32, copy=False)
    im_scales = [np.array(im_scale)]
    im_info = np.array([[data.shape[2], data.shape[3], im_scales[0]]], dtype=np.float32)
    net.blobs["data"].reshape(*data.shape)
    net.blobs["im_info"].reshape(*im_info.shape)
    return {"data": data, "im_info": im_info}, im_scales


def im_detect(im, net):
    """-> boxes [2R,4] (original-image pixels), masks [2R,1,21,21], seg scores [2R,21] of stages 3 and 5.
    With cfg.TEST.DEVICE_RESULTS (default) the three results stay on the GPU as DeviceArrays -- gpu_mask_voting consumes them
    there, np.asarray() / indexing gives the reference's numpy arrays."""
    forward_kwargs, im_scales = prepare_mnc_args(im, net)
    net.forward(**forward_kwargs)
    scale = np.float32(im_scales[0])      # float32 un-scaling: what numpy-1.x value-based casting did in the reference
    if cfg.TEST.get("DEVICE_RESULTS", True) and hasattr(net, "detect_tail"):
        return net.detect_tail(scale, im.shape)
    stage_boxes = []
    for name in ("rois", "rois_ext"):
        rois = net.blobs[name].data.copy()
        stage_boxes.append(clip_boxes(rois[:, 1:5] / scale, im.shape)[0])
    masks = np.concatenate((net.blobs["mask_proposal"].data, net.blobs["mask_proposal_ext"].data), axis=0)
    scores = np.concatenate((net.blobs["seg_cls_prob"].data, net.blobs["seg_cls_prob_ext"].data), axis=0)
    return np.concatenate(stage_boxes, axis=0), masks, scores


def get_vis_dict(result_box, result_mask, img_name, cls_names, vis_thresh=0.5):
    boxes, masks, classes = [], [], []
    for cls_ind in range(len(cls_names)):
        det, seg = result_box[cls_ind], result_mask[cls_ind]
        for k in np.where(det[:, -1] >= vis_thresh)[0]:
            boxes.append(det[k])
            masks.append(seg[k][0])
            classes.append(cls_ind + 1)
    return {"image_name": img_name, "cls_name": classes, "boxes": boxes, "masks": masks}


def _read_image_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def _visualise(im_bgr, pred, out_path):
    """The tail of the reference demo (tools/demo.py:150-191): class-id image of the voted instances
    (lib/utils/vis_seg.py:_convert_pred_to_image) in VOC colours, blended 0.8 over the photo, one "<class> <score>" label per
    instance at its box corner, saved as PNG (matplotlib; PIL-only without the labels when matplotlib is missing)."""
    from PIL import Image
    from utils.vis_seg import _convert_pred_to_image, _get_voc_color_map
    h, w = im_bgr.shape[:2]
    _, cls_img = _convert_pred_to_image(w, h, pred)
    cls_rgb = _get_voc_color_map().astype(np.uint8)[cls_img]
    background = Image.fromarray(np.ascontiguousarray(im_bgr[:, :, ::-1])).convert("RGBA")
    blended = Image.blend(background, Image.fromarray(cls_rgb).convert("RGBA"), 0.8).convert("RGB")
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        blended.save(out_path)
        return
    fig, ax = plt.subplots(figsize=(12, 12))
    ax.imshow(np.asarray(blended), aspect="equal")
    for box, cls_ind in zip(pred["boxes"], pred["cls_name"]):
        ax.text(box[0], box[1] - 8, "{:s} {:.4f}".format(CLASSES[cls_ind - 1], box[-1]), bbox=dict(facecolor="blue", alpha=0.5),
                fontsize=14, color="white")
    plt.axis("off")
    plt.tight_layout()
    fig.savefig(out_path)
    plt.close(fig)


def build_net(args):
    from mnc_amd import models, synth
    prototxt = args.prototxt or models.write_mnc_5stage_test_prototxt()
    if args.caffemodel:
        weights = args.caffemodel
    else:
        print("no --net given: using seeded synthetic weights (detections are meaningless, timing is not)")
        weights = synth.synthetic_weights(prototxt, seed=0)
    caffe.set_mode_gpu()
    caffe.set_device(args.gpu_id)
    cfg.GPU_ID = args.gpu_id
    return caffe.Net(prototxt, weights, caffe.TEST)


def main(argv=None):
    args = parse_args(argv)
    net = build_net(args)
    warm = 128 * np.ones((300, 500, 3), dtype=np.float32)
    for _ in range(2):
        im_detect(warm, net)
    images = args.images
    if images is None:
        demo_dir = os.path.join(cfg.DATA_DIR, "demo")
        images = sorted(os.path.join(demo_dir, f) for f in os.listdir(demo_dir)) if os.path.isdir(demo_dir) else []
    if not images:
        print("no images given; running one synthetic 600x1000 image")
        images = [None]
    for path in images:
        print("~" * 35)
        print("Demo for {}".format(path or "<synthetic 600x1000>"))
        im = _read_image_bgr(path) if path else np.random.default_rng(0).integers(0, 256, (600, 1000, 3), dtype=np.uint8)
        start = time.time()
        boxes, masks, seg_scores = im_detect(im, net)
        print("forward time %f" % (time.time() - start))
        start = time.time()
        result_mask, result_box = gpu_mask_voting(masks, boxes, seg_scores, len(CLASSES) + 1, 100, im.shape[1], im.shape[0])
        print("mask voting time %f" % (time.time() - start))
        pred = get_vis_dict(result_box, result_mask, path or "synthetic", CLASSES, args.vis_thresh)
        print("%d instances with score >= %g" % (len(pred["boxes"]), args.vis_thresh))
        if args.vis and path:
            out = os.path.splitext(path)[0] + "_mnc.png"
            if args.out_dir:
                os.makedirs(args.out_dir, exist_ok=True)
                out = os.path.join(args.out_dir, os.path.basename(out))
            _visualise(im, pred, out)
            print("wrote", out)
    net.close()


if __name__ == "__main__":
    main()
