"""Instance-segmentation evaluation, mAP^r of SDS (reference: lib/utils/voc_eval.py:19-52 voc_ap, :195-283 voc_eval_sds,
:307-352 parse_inst, :355-393 check_voc_sds_cache; lib/transform/mask_transform.py:16-46 mask_overlap).  Python-3 port of
the caller on the output side of the hot path: pickles are binary, dict iteration is .items(),
cv2.resize is utils.blob.resize_to."""
import os
import pickle

import numpy as np

from mnc_config import cfg
from transform.mask_transform import mask_overlap
from utils.blob import resize_to


def voc_ap(rec, prec, use_07_metric=False):
    """AP from recall/precision: the VOC07 11-point interpolation, or the area under the precision envelope."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def parse_rec(filename):
    """Objects of a PASCAL VOC annotation file: name, pose, truncated, difficult, bbox [xmin, ymin, xmax, ymax]."""
    import xml.etree.ElementTree as ET
    objects = []
    for obj in ET.parse(filename).findall('object'):
        bbox = obj.find('bndbox')
        objects.append({'name': obj.find('name').text, 'pose': obj.find('pose').text,
                        'truncated': int(obj.find('truncated').text), 'difficult': int(obj.find('difficult').text),
                        'bbox': [int(bbox.find('xmin').text), int(bbox.find('ymin').text), int(bbox.find('xmax').text),
                                 int(bbox.find('ymax').text)]})
    return objects


def voc_eval(detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=False):
    """PASCAL VOC detection AP of one class (reference: lib/utils/voc_eval.py:55-192): detpath.format(classname) is the
    results file (`image score x1 y1 x2 y2` per line), annopath.format(image) the XML annotation.  Returns rec, prec, ap."""
    if not os.path.isdir(cachedir):
        os.mkdir(cachedir)
    cachefile = os.path.join(cachedir, 'annots.pkl')
    with open(imagesetfile, 'r') as f:
        imagenames = [x.strip() for x in f.readlines()]
    if not os.path.isfile(cachefile):
        recs = {}
        for i, imagename in enumerate(imagenames):
            recs[imagename] = parse_rec(annopath.format(imagename))
            if i % 100 == 0:
                print('Reading annotation for {:d}/{:d}'.format(i + 1, len(imagenames)))
        print('Saving cached annotations to {:s}'.format(cachefile))
        with open(cachefile, 'wb') as f:
            pickle.dump(recs, f)
    else:
        with open(cachefile, 'rb') as f:
            recs = pickle.load(f)

    class_recs = {}
    npos = 0
    for imagename in imagenames:
        R = [obj for obj in recs[imagename] if obj['name'] == classname]
        bbox = np.array([x['bbox'] for x in R])
        difficult = np.array([x['difficult'] for x in R]).astype(bool)
        npos = npos + sum(~difficult)
        class_recs[imagename] = {'bbox': bbox, 'difficult': difficult, 'det': [False] * len(R)}

    with open(detpath.format(classname), 'r') as f:
        splitlines = [x.strip().split(' ') for x in f.readlines()]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines])

    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :] if len(sorted_ind) else BB
    image_ids = [image_ids[x] for x in sorted_ind]

    nd = len(image_ids)
    tp = np.zeros(nd)
    fp = np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        bb = BB[d, :].astype(float)
        ovmax = -np.inf
        BBGT = R['bbox'].astype(float)
        if BBGT.size > 0:
            ixmin = np.maximum(BBGT[:, 0], bb[0])
            iymin = np.maximum(BBGT[:, 1], bb[1])
            ixmax = np.minimum(BBGT[:, 2], bb[2])
            iymax = np.minimum(BBGT[:, 3], bb[3])
            iw = np.maximum(ixmax - ixmin + 1., 0.)
            ih = np.maximum(iymax - iymin + 1., 0.)
            inters = iw * ih
            uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) +
                   (BBGT[:, 2] - BBGT[:, 0] + 1.) * (BBGT[:, 3] - BBGT[:, 1] + 1.) - inters)
            overlaps = inters / uni
            ovmax = np.max(overlaps)
            jmax = np.argmax(overlaps)
        if ovmax > ovthresh:
            if not R['difficult'][jmax]:
                if not R['det'][jmax]:
                    tp[d] = 1.
                    R['det'][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.

    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / (tp + fp)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def parse_inst(image_name, devkit_path):
    """Ground-truth instances of one SBD image: inst/<name>.mat (instance ids) + cls/<name>.mat (class ids) ->
    [{'mask': bool [h,w] inside its bounds, 'mask_cls': class id, 'mask_bound': [x1,y1,x2,y2]}]."""
    import scipy.io as sio
    inst = sio.loadmat(os.path.join(devkit_path, 'inst', image_name + '.mat'))['GTinst']['Segmentation'][0][0]
    cls = sio.loadmat(os.path.join(devkit_path, 'cls', image_name + '.mat'))['GTcls']['Segmentation'][0][0]
    record = []
    for inst_id in np.unique(inst):
        if inst_id == 0:                # background
            continue
        r, c = np.where(inst == inst_id)
        bound = np.array([np.min(c), np.min(r), np.max(c), np.max(r)], dtype=np.float64)
        x1, y1, x2, y2 = (int(v) for v in bound)
        mask = inst[y1:y2 + 1, x1:x2 + 1] == inst_id
        classes = np.unique(cls[y1:y2 + 1, x1:x2 + 1][mask])
        assert classes.shape[0] == 1
        record.append({'mask': mask, 'mask_cls': classes[0], 'mask_bound': bound})
    return record


def check_voc_sds_cache(cache_dir, devkit_path, image_names, class_names):
    """Builds <cache_dir>/<class>_mask_gt.pkl ({image: [instances]}) once."""
    if not os.path.isdir(cache_dir):
        os.mkdir(cache_dir)
    if all(os.path.isfile(os.path.join(cache_dir, n + '_mask_gt.pkl')) for n in class_names if n != '__background__'):
        return
    record_list = [{} for _ in range(len(class_names))]
    for i, image_name in enumerate(image_names):
        for mask_dic in parse_inst(image_name, devkit_path):
            mask_dic['already_detect'] = False
            record_list[int(mask_dic['mask_cls'])].setdefault(image_name, []).append(mask_dic)
        if i % 100 == 0:
            print('Reading annotation for {:d}/{:d}'.format(i + 1, len(image_names)))
    print('Saving cached annotations...')
    for cls_ind, name in enumerate(class_names):
        if name == '__background__':
            continue
        with open(os.path.join(cache_dir, name + '_mask_gt.pkl'), 'wb') as f:
            pickle.dump(record_list[cls_ind], f)


def voc_eval_sds(det_file, seg_file, devkit_path, image_list, cls_name, cache_dir, class_names, ov_thresh=0.5):
    """AP^r of one class: predictions ranked by score, a prediction is a true positive when its mask (21x21 resized to its
    box, binarised at cfg.BINARIZE_THRESH) overlaps a not-yet-matched ground-truth instance of the class by >= ov_thresh."""
    with open(image_list, 'r') as f:
        image_names = [x.strip() for x in f.readlines()]
    check_voc_sds_cache(cache_dir, devkit_path, image_names, class_names)
    with open(cache_dir + '/' + cls_name + '_mask_gt.pkl', 'rb') as f:
        gt_pkl = pickle.load(f)
    with open(det_file, 'rb') as f:
        boxes_pkl = pickle.load(f)
    with open(seg_file, 'rb') as f:
        masks_pkl = pickle.load(f)

    box_num = sum(len(boxes_pkl[i]) for i in range(len(image_names)))
    new_boxes = np.zeros((box_num, 5))
    new_masks = np.zeros((box_num, cfg.MASK_SIZE, cfg.MASK_SIZE))
    new_image = []
    cnt = 0
    for image_ind, name in enumerate(image_names):
        boxes, masks = boxes_pkl[image_ind], masks_pkl[image_ind]
        for box_ind in range(len(boxes)):
            new_boxes[cnt] = boxes[box_ind]
            new_masks[cnt] = masks[box_ind]
            new_image.append(name)
            cnt += 1

    keep_inds = np.argsort(-new_boxes[:, -1])
    new_boxes = new_boxes[keep_inds, :]
    new_masks = new_masks[keep_inds, :, :]
    num_pred = new_boxes.shape[0]

    fp = np.zeros((num_pred, 1))
    tp = np.zeros((num_pred, 1))
    for i in range(num_pred):
        pred_box = np.round(new_boxes[i, :4]).astype(int)
        pred_mask = resize_to(new_masks[i].astype(np.float32), pred_box[2] - pred_box[0] + 1, pred_box[3] - pred_box[1] + 1)
        pred_mask = pred_mask >= cfg.BINARIZE_THRESH
        image_index = new_image[keep_inds[i]]
        if image_index not in gt_pkl:
            fp[i] = 1
            continue
        gt_dict_list = gt_pkl[image_index]
        cur_overlap, cur_overlap_ind = -1000, -1
        for ind2, gt_dict in enumerate(gt_dict_list):
            ov = mask_overlap(np.round(gt_dict['mask_bound']).astype(int), pred_box, gt_dict['mask'], pred_mask)
            if ov > cur_overlap:
                cur_overlap, cur_overlap_ind = ov, ind2
        if cur_overlap >= ov_thresh and not gt_dict_list[cur_overlap_ind]['already_detect']:
            tp[i] = 1
            gt_dict_list[cur_overlap_ind]['already_detect'] = 1
        else:
            fp[i] = 1

    num_pos = sum(len(val) for val in gt_pkl.values())
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(num_pos)
    prec = tp / np.maximum(fp + tp, np.finfo(np.float64).eps)
    return voc_ap(rec, prec, True)
