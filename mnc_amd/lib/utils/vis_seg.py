"""Rendering of instance-segmentation results (reference: lib/utils/vis_seg.py:14-147; the tail of tools/demo.py:150-191).
SURVEY section 8f row n4 (visualisation tail).  Host-only; cv2 is replaced by numpy / PIL:
`_convert_pred_to_image` and `_get_voc_color_map` are pinned against the reference's functions
(tests/golden/make_golden_eval.py); file encoding (JPEG / PNG bytes) is PIL's, not OpenCV's."""
import os
import pickle

import numpy as np

from mnc_config import cfg
from utils.blob import resize_to
from utils.image_io import imread


def vis_seg(img_names, cls_names, output_dir, gt_dir, image_ext='.jpg'):
    """For every image: SegInst/<name>.jpg (instance ids in VOC colours), SegCls/<name>.jpg (class ids + box outlines) and
    SegRes/<name>.png (the class image blended 0.8 over the photo), from output_dir/res_boxes.pkl + res_masks.pkl."""
    from PIL import Image
    assert os.path.exists(output_dir)
    inst_dir, cls_dir, res_dir = (os.path.join(output_dir, d) for d in ('SegInst', 'SegCls', 'SegRes'))
    for d in (inst_dir, cls_dir, res_dir):
        if not os.path.isdir(d):
            os.mkdir(d)
    res_list = _prepare_dict(img_names, cls_names, output_dir)
    color_map = _get_voc_color_map().astype(np.uint8)
    for img_ind, image_name in enumerate(img_names):
        print(image_name)
        img_data = imread(os.path.join(gt_dir, 'img', image_name + image_ext))            # BGR
        img_height, img_width = img_data.shape[:2]
        inst_img, cls_img = _convert_pred_to_image(img_width, img_height, res_list[img_ind])
        inst_rgb, cls_rgb = color_map[inst_img], color_map[cls_img]                       # cv2.imwrite of the BGR-flipped map
        Image.fromarray(inst_rgb).save(os.path.join(inst_dir, image_name + '.jpg'))
        Image.fromarray(cls_rgb).save(os.path.join(cls_dir, image_name + '.jpg'))
        background = Image.fromarray(np.ascontiguousarray(img_data[:, :, ::-1])).convert('RGBA')
        blended = Image.blend(background, Image.fromarray(cls_rgb).convert('RGBA'), 0.8)
        blended.save(os.path.join(res_dir, image_name + '.png'), 'PNG')


def _prepare_dict(img_names, cls_names, cache_dir, vis_thresh=0.5):
    """Per image {'image_name', 'cls_name': [class index], 'boxes': [[x1,y1,x2,y2,score]], 'masks': [21x21]} of the
    detections scoring >= vis_thresh (vis_seg.py:64-98)."""
    with open(os.path.join(cache_dir, 'res_boxes.pkl'), 'rb') as f:
        det_pkl = pickle.load(f)
    with open(os.path.join(cache_dir, 'res_masks.pkl'), 'rb') as f:
        seg_pkl = pickle.load(f)
    res_list = []
    for img_ind, image_name in enumerate(img_names):
        box_for_img, mask_for_img, cls_for_img = [], [], []
        for cls_ind, cls_name in enumerate(cls_names):
            if cls_name == '__background__' or len(det_pkl[cls_ind][img_ind]) == 0:
                continue
            det_for_img, seg_for_img = det_pkl[cls_ind][img_ind], seg_pkl[cls_ind][img_ind]
            for keep in np.where(det_for_img[:, -1] >= vis_thresh)[0]:
                box_for_img.append(det_for_img[keep])
                mask_for_img.append(seg_for_img[keep][0])
                cls_for_img.append(cls_ind)
        res_list.append({'image_name': image_name, 'cls_name': cls_for_img, 'boxes': box_for_img, 'masks': mask_for_img})
    return res_list


def _convert_pred_to_image(img_width, img_height, pred_dict):
    """Instance-id image and class-id image (vis_seg.py:101-131): each mask is resized to its (rounded, clipped) box,
    binarised at cfg.BINARIZE_THRESH and painted in order -- later instances overwrite earlier ones -- and the class image
    gets 2-pixel box outlines of value 150 (slices that start at -1 are empty, as in the reference)."""
    num_inst = len(pred_dict['boxes'])
    inst_img = np.zeros((img_height, img_width))
    cls_img = np.zeros((img_height, img_width))
    for i in range(num_inst):
        box = np.round(pred_dict['boxes'][i]).astype(int)
        mask = pred_dict['masks'][i]
        cls_num = pred_dict['cls_name'][i]
        box[0] = min(max(box[0], 0), img_width - 1)
        box[1] = min(max(box[1], 0), img_height - 1)
        box[2] = min(max(box[2], 0), img_width - 1)
        box[3] = min(max(box[3], 0), img_height - 1)
        mask = resize_to(mask.astype(np.float32), box[2] - box[0] + 1, box[3] - box[1] + 1)
        mask = mask >= cfg.BINARIZE_THRESH
        keep = np.logical_not(mask)
        part1 = (i + 1) * mask.astype(np.float32)
        part2 = np.multiply(keep, inst_img[box[1]:box[3] + 1, box[0]:box[2] + 1])
        part3 = np.multiply(keep, cls_img[box[1]:box[3] + 1, box[0]:box[2] + 1])
        inst_img[box[1]:box[3] + 1, box[0]:box[2] + 1] = part1 + part2
        cls_img[box[1]:box[3] + 1, box[0]:box[2] + 1] = cls_num * mask.astype(np.float32) + part3
        cls_img[box[1]:box[3] + 1, box[0] - 1:box[0] + 1] = 150
        cls_img[box[1]:box[3] + 1, box[2] - 1:box[2] + 1] = 150
        cls_img[box[1] - 1:box[1] + 1, box[0]:box[2] + 1] = 150
        cls_img[box[3] - 1:box[3] + 1, box[0]:box[2] + 1] = 150
    return inst_img.astype(int), cls_img.astype(int)


def _get_voc_color_map(n=256):
    """The PASCAL VOC label colour map (vis_seg.py:134-147): bit k of (r, g, b) from bits (3j, 3j+1, 3j+2) of the label."""
    color_map = np.zeros((n, 3))
    for i in range(n):
        r = g = b = 0
        cid = i
        for j in range(8):
            r |= ((cid >> 0) & 1) << (7 - j)
            g |= ((cid >> 1) & 1) << (7 - j)
            b |= ((cid >> 2) & 1) << (7 - j)
            cid >>= 3
        color_map[i] = (r, g, b)
    return color_map
