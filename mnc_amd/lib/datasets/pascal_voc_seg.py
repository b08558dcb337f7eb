"""PASCAL VOC / SBD instance-segmentation image database, test-time surface (reference:
lib/datasets/pascal_voc_seg.py:19-44,155-228 and the parts of pascal_voc_det.py / db/imdb.py it inherits that
tools/test_net.py --task seg / vis_seg touches: name, classes, image_index, image_path_at, evaluate_segmentation,
visualization_segmentation).

Layout of the devkit (data/VOCdevkitSDS): img/<id>.jpg, inst/<id>.mat, cls/<id>.mat, <image_set>.txt.
Training-time members (roidb / maskdb construction, flipping) are outside the inference hot path and not provided."""
import os
import pickle

import numpy as np

from mnc_config import cfg
from utils.voc_eval import voc_eval_sds

CLASSES = ('__background__',  # always index 0
           'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
           'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


class PascalVOCSeg(object):
    def __init__(self, image_set, year, devkit_path=None, image_ext='.jpg'):
        self._name = 'voc_' + year + '_' + image_set
        self._year = year
        self._image_set = image_set
        self._devkit_path = os.path.join(cfg.DATA_DIR, 'VOCdevkitSDS') if devkit_path is None else devkit_path
        if not os.path.isabs(self._devkit_path):
            self._devkit_path = os.path.join(cfg.ROOT_DIR, self._devkit_path)
        self._data_path = self._devkit_path
        self._classes = CLASSES
        self._image_ext = image_ext
        assert os.path.exists(self._devkit_path), 'VOCdevkit path does not exist: {}'.format(self._devkit_path)
        self._image_index = self._load_image_set_index()

    name = property(lambda self: self._name)
    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)
    num_images = property(lambda self: len(self._image_index))

    def image_path_at(self, i):
        image_path = os.path.join(self._data_path, 'img', self._image_index[i] + self._image_ext)
        assert os.path.exists(image_path), 'Path does not exist: {}'.format(image_path)
        return image_path

    def _load_image_set_index(self):
        image_set_file = os.path.join(self._data_path, self._image_set + '.txt')
        assert os.path.exists(image_set_file), 'Path does not exist: {}'.format(image_set_file)
        with open(image_set_file) as f:
            return [x.strip() for x in f.readlines()]

    # --------------------------- Evaluation ---------------------------
    def visualization_segmentation(self, output_dir):
        """pascal_voc_seg.py:152-153: render res_boxes.pkl / res_masks.pkl of `output_dir` over the images."""
        from utils.vis_seg import vis_seg
        vis_seg(self.image_index, self.classes, output_dir, self._data_path, self._image_ext)

    def evaluate_segmentation(self, all_boxes, all_masks, output_dir):
        self._write_voc_seg_results_file(all_boxes, all_masks, output_dir)
        return self._py_evaluate_segmentation(output_dir)

    def _write_voc_seg_results_file(self, all_boxes, all_masks, output_dir):
        """<class>_det.pkl = all_boxes[cls] ([n,5] per image), <class>_seg.pkl = binarised [n,21,21] masks per image."""
        all_boxes, all_masks = self._reformat_result(all_boxes, all_masks)
        for cls_inds, cls in enumerate(self.classes):
            if cls == '__background__':
                continue
            print('Writing {} VOC results file'.format(cls))
            with open(os.path.join(output_dir, cls + '_det.pkl'), 'wb') as f:
                pickle.dump(all_boxes[cls_inds], f, pickle.HIGHEST_PROTOCOL)
            with open(os.path.join(output_dir, cls + '_seg.pkl'), 'wb') as f:
                pickle.dump(all_masks[cls_inds], f, pickle.HIGHEST_PROTOCOL)

    def _reformat_result(self, boxes, masks):
        num_images = len(self.image_index)
        out = [[[] for _ in range(num_images)] for _ in range(len(self.classes))]
        for c in range(1, len(self.classes)):
            for i in range(num_images):
                if len(masks[c][i]) == 0:
                    continue
                m = np.asarray(masks[c][i])
                out[c][i] = m.reshape(m.shape[0], cfg.MASK_SIZE, cfg.MASK_SIZE) >= cfg.BINARIZE_THRESH
        return boxes, out

    def _py_evaluate_segmentation(self, output_dir):
        gt_dir = self._data_path
        imageset_file = os.path.join(gt_dir, self._image_set + '.txt')
        cache_dir = os.path.join(self._devkit_path, 'annotations_cache')
        if not os.path.isdir(output_dir):
            os.mkdir(output_dir)
        print('VOC07 metric? Yes')                      # SDS's evaluation protocol
        result = {}
        for thr in (0.5, 0.7):
            print('~~~~~~ Evaluation use min overlap = {} ~~~~~~'.format(thr))
            aps = []
            for cls in self._classes:
                if cls == '__background__':
                    continue
                ap = voc_eval_sds(os.path.join(output_dir, cls + '_det.pkl'), os.path.join(output_dir, cls + '_seg.pkl'),
                                  gt_dir, imageset_file, cls, cache_dir, self._classes, ov_thresh=thr)
                aps.append(ap)
                print('AP for {} = {:.2f}'.format(cls, ap * 100))
            print('Mean AP@{} = {:.2f}'.format(thr, np.mean(aps) * 100))
            result[thr] = aps
        return result
