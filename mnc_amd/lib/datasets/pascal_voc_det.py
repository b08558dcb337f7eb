"""PASCAL VOC detection image database, test-time surface (reference: lib/datasets/pascal_voc_det.py:20-70, 225-305 and the
db/imdb.py members `tools/test_net.py --task det` touches: name, classes, image_index, image_path_at, evaluate_detections).

Layout of the devkit (data/VOCdevkit2007): VOC2007/JPEGImages/<id>.jpg, VOC2007/Annotations/<id>.xml,
VOC2007/ImageSets/Main/<set>.txt, results/VOC2007/Main/ (written).  Training-time members (roidb construction) are outside
the inference path and not provided."""
import os
import pickle
import uuid

import numpy as np

from datasets.pascal_voc_seg import CLASSES
from mnc_config import cfg
from utils.voc_eval import voc_eval


class PascalVOCDet(object):
    def __init__(self, image_set, year, devkit_path=None, image_ext='.jpg'):
        self._name = 'voc_' + year + '_' + image_set
        self._year = year
        self._image_set = image_set
        self._devkit_path = os.path.join(cfg.DATA_DIR, 'VOCdevkit' + year) if devkit_path is None else devkit_path
        if not os.path.isabs(self._devkit_path):
            self._devkit_path = os.path.join(cfg.ROOT_DIR, self._devkit_path)
        self._data_path = os.path.join(self._devkit_path, 'VOC' + self._year)
        self._classes = CLASSES
        self._image_ext = image_ext
        self._salt = str(uuid.uuid4())
        self._comp_id = 'comp4'
        self.config = {'cleanup': True, 'use_salt': True, 'matlab_eval': False}
        assert os.path.exists(self._devkit_path), 'VOCdevkit path does not exist: {}'.format(self._devkit_path)
        assert os.path.exists(self._data_path), 'Path does not exist: {}'.format(self._data_path)
        self._image_index = self._load_image_set_index()

    name = property(lambda self: self._name)
    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)
    num_images = property(lambda self: len(self._image_index))

    def image_path_at(self, i):
        image_path = os.path.join(self._data_path, 'JPEGImages', self._image_index[i] + self._image_ext)
        assert os.path.exists(image_path), 'Path does not exist: {}'.format(image_path)
        return image_path

    def _load_image_set_index(self):
        image_set_file = os.path.join(self._data_path, 'ImageSets', 'Main', self._image_set + '.txt')
        assert os.path.exists(image_set_file), 'Path does not exist: {}'.format(image_set_file)
        with open(image_set_file) as f:
            return [x.strip() for x in f.readlines()]

    # --------------------------- Evaluation ---------------------------
    def evaluate_detections(self, all_boxes, output_dir):
        self._write_voc_results_file(all_boxes)
        aps = self._do_python_eval(output_dir)
        if self.config['matlab_eval']:
            raise NotImplementedError
        if self.config['cleanup']:
            for cls in self._classes:
                if cls != '__background__':
                    os.remove(self._get_voc_results_file_template().format(cls))
        return aps

    def _get_comp_id(self):
        return self._comp_id + '_' + self._salt if self.config['use_salt'] else self._comp_id

    def _get_voc_results_file_template(self):
        # VOCdevkit/results/VOC2007/Main/<comp_id>_det_test_aeroplane.txt
        filename = self._get_comp_id() + '_det_' + self._image_set + '_{:s}.txt'
        return os.path.join(self._devkit_path, 'results', 'VOC' + self._year, 'Main', filename)

    def _write_voc_results_file(self, all_boxes):
        os.makedirs(os.path.dirname(self._get_voc_results_file_template()), exist_ok=True)
        for cls_ind, cls in enumerate(self.classes):
            if cls == '__background__':
                continue
            print('Writing {} VOC results file'.format(cls))
            with open(self._get_voc_results_file_template().format(cls), 'wt') as f:
                for im_ind, index in enumerate(self.image_index):
                    dets = all_boxes[cls_ind][im_ind]
                    if len(dets) == 0:
                        continue
                    for k in range(dets.shape[0]):          # the VOCdevkit expects 1-based pixel indices
                        f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(
                            index, dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))

    def _do_python_eval(self, output_dir='output'):
        annopath = os.path.join(self._devkit_path, 'VOC' + self._year, 'Annotations', '{:s}.xml')
        imagesetfile = os.path.join(self._devkit_path, 'VOC' + self._year, 'ImageSets', 'Main', self._image_set + '.txt')
        cachedir = os.path.join(self._devkit_path, 'annotations_cache')
        use_07_metric = int(self._year) < 2010          # the PASCAL VOC metric changed in 2010
        print('VOC07 metric? ' + ('Yes' if use_07_metric else 'No'))
        if not os.path.isdir(output_dir):
            os.mkdir(output_dir)
        aps = []
        for cls in self._classes:
            if cls == '__background__':
                continue
            rec, prec, ap = voc_eval(self._get_voc_results_file_template().format(cls), annopath, imagesetfile, cls, cachedir,
                                     ovthresh=0.5, use_07_metric=use_07_metric)
            aps.append(ap)
            print('AP for {} = {:.4f}'.format(cls, ap))
            with open(os.path.join(output_dir, cls + '_pr.pkl'), 'wb') as f:
                pickle.dump({'rec': rec, 'prec': prec, 'ap': ap}, f)
        print('Mean AP = {:.4f}'.format(np.mean(aps)))
        return aps
