"""Test-time driver of the reference's three inference graphs over an image database (reference:
lib/caffeWrapper/TesterWrapper.py:25-414).  `seg` = MNC 5-stage: per-image forward -> un-scale/clip/concat of both stages ->
gpu_mask_voting (or per-class NMS) -> res_boxes.pkl / res_masks.pkl -> imdb.evaluate_segmentation (SURVEY section 8f row
n1); `det` = Faster R-CNN end2end and `cfm` = convolutional feature masking over MCG proposals (row n3); `vis_seg` renders
the stored results (row n4's visualisation tail)."""
import heapq
import os
import pickle

import numpy as np

import caffe
from mnc_config import cfg, get_output_dir
from nms.nms_wrapper import apply_nms, apply_nms_mask_single
from transform.bbox_transform import bbox_transform_inv, clip_boxes, filter_small_boxes
from transform.mask_transform import gpu_mask_voting
from utils.blob import (can_prep_on_device, cfm_scale_factors, im_list_to_blob, prep_im_for_blob, prep_im_for_blob_cfm,
                        prep_im_for_blob_cfm_device, prep_im_for_blob_device, pred_rois_for_blob, resize_to)
from utils.image_io import imread
from utils.timer import Timer


class TesterWrapper(object):
    def __init__(self, test_prototxt, imdb, test_model, task_name):
        self.net = caffe.Net(test_prototxt, test_model, caffe.TEST)
        self.net.name = os.path.splitext(os.path.basename(str(test_model)))[0] if not isinstance(test_model, dict) \
            else self.net.name
        self.imdb = imdb
        self.output_dir = get_output_dir(imdb, self.net)
        self.task_name = task_name
        self.num_images = len(self.imdb.image_index)
        self.num_classes = self.imdb.num_classes
        self.max_per_set = 40 * self.num_images       # heuristic: 40 detections per class per image before NMS
        self.max_per_image = 100                      # heuristic: at most 100 detections per class per image
        if not os.path.exists(self.output_dir):
            os.makedirs(self.output_dir)

    def get_result(self):
        det_file = os.path.join(self.output_dir, 'res_boxes.pkl')
        seg_file = os.path.join(self.output_dir, 'res_masks.pkl')
        if self.task_name == 'det':
            return self.get_detection_result()
        if self.task_name == 'vis_seg':
            return self.vis_segmentation_result()
        if self.task_name not in ('seg', 'cfm'):
            print("task name only support 'det', 'seg', 'cfm' and 'vis_seg'")
            raise NotImplementedError(self.task_name)
        if os.path.isfile(det_file) and os.path.isfile(seg_file):
            with open(det_file, 'rb') as f:
                seg_box = pickle.load(f)
            with open(seg_file, 'rb') as f:
                seg_mask = pickle.load(f)
        else:
            seg_box, seg_mask = self.get_segmentation_result() if self.task_name == 'seg' else self.get_cfm_result()
            with open(det_file, 'wb') as f:
                pickle.dump(seg_box, f, pickle.HIGHEST_PROTOCOL)
            with open(seg_file, 'wb') as f:
                pickle.dump(seg_mask, f, pickle.HIGHEST_PROTOCOL)
        print('Evaluating segmentation using MNC 5 stage inference' if self.task_name == 'seg' else
              'Evaluating segmentation using convolutional feature masking')
        return self.imdb.evaluate_segmentation(seg_box, seg_mask, self.output_dir)

    def vis_segmentation_result(self):
        """TesterWrapper.py:146-147: render the result pickles a previous `seg` / `cfm` run left in output_dir."""
        return self.imdb.visualization_segmentation(self.output_dir)

    def get_detection_result(self):
        """Faster R-CNN end2end test loop (TesterWrapper.py:86-143): all_boxes[cls][image] = [n,5], per-class score
        thresholds from the max_per_set heap, detections.pkl, per-class NMS, imdb.evaluate_detections."""
        max_per_set = 40 * self.num_images
        max_per_image = 100
        thresh = -np.inf * np.ones(self.num_classes)
        top_scores = [[] for _ in range(self.num_classes)]
        all_boxes = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        _t = {'im_detect': Timer(), 'misc': Timer()}
        for i in range(self.num_images):
            im = imread(self.imdb.image_path_at(i))
            _t['im_detect'].tic()
            scores, boxes = self._detection_forward(im)
            _t['im_detect'].toc()
            for j in range(1, self.num_classes):
                inds = np.where(scores[:, j] > thresh[j])[0]
                cls_scores = scores[inds, j]
                cls_boxes = boxes[inds, j * 4:(j + 1) * 4]
                top_inds = np.argsort(-cls_scores)[:max_per_image]
                cls_scores, cls_boxes = cls_scores[top_inds], cls_boxes[top_inds, :]
                for val in cls_scores:
                    heapq.heappush(top_scores[j], val)
                if len(top_scores[j]) > max_per_set:
                    while len(top_scores[j]) > max_per_set:
                        heapq.heappop(top_scores[j])
                    thresh[j] = top_scores[j][0]
                all_boxes[j][i] = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
            print('process image %d/%d, forward average time %f' % (i, self.num_images, _t['im_detect'].average_time))
        for j in range(1, self.num_classes):
            for i in range(self.num_images):
                inds = np.where(all_boxes[j][i][:, -1] > thresh[j])[0]
                all_boxes[j][i] = all_boxes[j][i][inds, :]
        with open(os.path.join(self.output_dir, 'detections.pkl'), 'wb') as f:
            pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
        print('Applying NMS to all detections')
        nms_dets = apply_nms(all_boxes, cfg.TEST.NMS)
        print('Evaluating detections')
        return self.imdb.evaluate_detections(nms_dets, self.output_dir)

    def _detection_forward(self, im):
        """-> scores [R,K] and boxes [R,4K] (per-class regressed boxes in original-image pixels), TesterWrapper.py:216-238."""
        forward_kwargs, im_scales = self._prepare_mnc_args(im)
        blobs_out = self.net.forward(**forward_kwargs)
        rois = self.net.blobs['rois'].data.copy()
        boxes = rois[:, 1:5] / np.float32(im_scales[0])       # un-scale back to raw image space (float32, as numpy 1.x)
        pred_boxes = bbox_transform_inv(boxes, blobs_out['bbox_pred'])
        pred_boxes, _ = clip_boxes(pred_boxes, im.shape)
        return blobs_out['cls_prob'], pred_boxes

    def get_segmentation_result(self):
        """all_boxes[cls][image] = [n,5] (x1,y1,x2,y2,score), all_masks[cls][image] = [n,1,21,21] float."""
        thresh = -np.inf * np.ones(self.num_classes)          # adaptively raised by the max_per_set constraint
        top_scores = [[] for _ in range(self.num_classes)]    # one min-heap of scores per class
        all_boxes = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        all_masks = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        _t = {'im_detect': Timer(), 'misc': Timer()}
        for i in range(self.num_images):
            im = imread(self.imdb.image_path_at(i))
            _t['im_detect'].tic()
            masks, boxes, seg_scores = self._segmentation_forward(im)
            _t['im_detect'].toc()
            if not cfg.TEST.USE_MASK_MERGE:
                for j in range(1, self.num_classes):
                    inds = np.where(seg_scores[:, j] > thresh[j])[0]
                    cls_scores, cls_boxes, cls_masks = seg_scores[inds, j], boxes[inds, :], masks[inds, :]
                    top_inds = np.argsort(-cls_scores)[:self.max_per_image]
                    cls_scores, cls_boxes, cls_masks = cls_scores[top_inds], cls_boxes[top_inds, :], cls_masks[top_inds, :]
                    for val in cls_scores:
                        heapq.heappush(top_scores[j], val)
                    if len(top_scores[j]) > self.max_per_set:
                        while len(top_scores[j]) > self.max_per_set:
                            heapq.heappop(top_scores[j])
                        thresh[j] = top_scores[j][0]
                    box_before_nms = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
                    mask_before_nms = cls_masks.astype(np.float32, copy=False)
                    all_boxes[j][i], all_masks[j][i] = apply_nms_mask_single(box_before_nms, mask_before_nms, cfg.TEST.NMS)
            else:
                if not cfg.TEST.USE_GPU_MASK_MERGE:
                    # the reference's cpu_mask_voting (mask_transform.py:142-211) is its CPU-only alternative; this
                    # package has no CPU compute path
                    raise NotImplementedError("cfg.TEST.USE_GPU_MASK_MERGE=False (cpu_mask_voting) is not provided")
                result_mask, result_box = gpu_mask_voting(masks, boxes, seg_scores, self.num_classes,
                                                          self.max_per_image, im.shape[1], im.shape[0])
                for j in range(1, self.num_classes):      # no heap: voting never returns more than max_per_image
                    all_boxes[j][i] = result_box[j - 1]
                    all_masks[j][i] = result_mask[j - 1]
            print('process image %d/%d, forward average time %f' % (i, self.num_images, _t['im_detect'].average_time))

        for j in range(1, self.num_classes):
            for i in range(self.num_images):
                inds = np.where(all_boxes[j][i][:, -1] > thresh[j])[0]
                all_boxes[j][i] = all_boxes[j][i][inds, :]
                all_masks[j][i] = all_masks[j][i][inds]
        return all_boxes, all_masks

    def _segmentation_forward(self, im):
        forward_kwargs, im_scales = self._prepare_mnc_args(im)
        self.net.forward(**forward_kwargs)
        if cfg.TEST.get("DEVICE_RESULTS", True) and hasattr(self.net, "detect_tail"):
            # same three results, left on the GPU for gpu_mask_voting (np.asarray() of them is the numpy path's output)
            boxes, masks, scores = self.net.detect_tail(np.float32(im_scales[0]), im.shape)
            return masks, boxes, scores
        rois_phase1 = self.net.blobs['rois'].data.copy()
        masks_phase1 = self.net.blobs['mask_proposal'].data[...]
        scores_phase1 = self.net.blobs['seg_cls_prob'].data[...]
        rois_phase2 = self.net.blobs['rois_ext'].data[...]
        masks_phase2 = self.net.blobs['mask_proposal_ext'].data[...]
        scores_phase2 = self.net.blobs['seg_cls_prob_ext'].data[...]
        # boxes are in the resized image's coordinates: un-scale, clip to the original image
        scale = np.float32(im_scales[0])      # float32 un-scaling: numpy-1.x value-based casting, as the reference ran
        rois_phase1, _ = clip_boxes(rois_phase1[:, 1:5] / scale, im.shape)
        rois_phase2, _ = clip_boxes(rois_phase2[:, 1:5] / scale, im.shape)
        masks = np.concatenate((masks_phase1, masks_phase2), axis=0)
        boxes = np.concatenate((rois_phase1, rois_phase2), axis=0)
        scores = np.concatenate((scores_phase1, scores_phase2), axis=0)
        return masks, boxes, scores

    def _prepare_mnc_args(self, im):
        if can_prep_on_device(self.net, im):         # mean subtraction + resize on the GPU; `data` stays there
            data, im_scale_factors = prep_im_for_blob_device(self.net, im, cfg.PIXEL_MEANS, cfg.TEST.SCALES[0],
                                                             cfg.TRAIN.MAX_SIZE)
        else:
            im, im_scale_factors = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TEST.SCALES[0], cfg.TRAIN.MAX_SIZE)
            data = im_list_to_blob([im]).astype(np.float32, copy=False)
        im_scales = [np.array(im_scale_factors)]
        im_info = np.array([[data.shape[2], data.shape[3], im_scales[0]]], dtype=np.float32)
        self.net.blobs['data'].reshape(*data.shape)
        self.net.blobs['im_info'].reshape(*im_info.shape)
        return {'data': data, 'im_info': im_info.astype(np.float32, copy=False)}, im_scales

    # ------------------------------------------------------------------------------------------------ CFM (row n3)
    def get_cfm_result(self):
        """TesterWrapper.py:286-335: per class score threshold heap, top max_per_image rows, NMS on boxes with the masks
        riding along; all_boxes[cls][image] = [n,5], all_masks[cls][image] = [n,1,21,21]."""
        thresh = -np.inf * np.ones(self.num_classes)
        top_scores = [[] for _ in range(self.num_classes)]
        all_boxes = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        all_masks = [[[] for _ in range(self.num_images)] for _ in range(self.num_classes)]
        _t = {'im_detect': Timer(), 'misc': Timer()}
        for i in range(self.num_images):
            _t['im_detect'].tic()
            masks, boxes, seg_scores = self.cfm_network_forward(i)
            for j in range(1, self.num_classes):
                inds = np.where(seg_scores[:, j] > thresh[j])[0]
                cls_scores, cls_boxes, cls_masks = seg_scores[inds, j], boxes[inds, :], masks[inds, :]
                top_inds = np.argsort(-cls_scores)[:self.max_per_image]
                cls_scores, cls_boxes, cls_masks = cls_scores[top_inds], cls_boxes[top_inds, :], cls_masks[top_inds, :]
                for val in cls_scores:
                    heapq.heappush(top_scores[j], val)
                if len(top_scores[j]) > self.max_per_set:
                    while len(top_scores[j]) > self.max_per_set:
                        heapq.heappop(top_scores[j])
                    thresh[j] = top_scores[j][0]
                box_before_nms = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
                mask_before_nms = cls_masks.astype(np.float32, copy=False)
                all_boxes[j][i], all_masks[j][i] = apply_nms_mask_single(box_before_nms, mask_before_nms, cfg.TEST.NMS)
            _t['im_detect'].toc()
            print('process image %d/%d, forward average time %f' % (i, self.num_images, _t['im_detect'].average_time))
        for j in range(1, self.num_classes):
            for i in range(self.num_images):
                inds = np.where(all_boxes[j][i][:, -1] > thresh[j])[0]
                all_boxes[j][i] = all_boxes[j][i][inds, :]
                all_masks[j][i] = all_masks[j][i][inds]
        return all_boxes, all_masks

    def _load_mcg_maskdb(self, im_i):
        """{'boxes': [n,4], 'masks': [n,S,S]} of image im_i: the .mat files tools/prepare_mcg_maskdb.py writes
        (TesterWrapper.py:339-341).  scipy is needed for this task only."""
        import scipy.io
        path = os.path.join(cfg.TEST.get('MCG_MASKDB_DIR', 'data/cache/voc_2012_val_mcg_maskdb/'),
                            self.imdb._image_index[im_i] + '.mat')
        return scipy.io.loadmat(path)

    def cfm_network_forward(self, im_i):
        """TesterWrapper.py:337-414: MCG proposals -> pyramid level per box (area closest to 224^2) -> per group of
        cfg.TEST.GROUP_SCALE adjacent levels one data blob, rois fed in chunks of cfg.TEST.MAX_ROIS_GPU[group] ->
        (mask_prob [n,1,21,21], the proposals' own boxes [n,4], seg_cls_prob [n,K]) in level-group order."""
        im = imread(self.imdb.image_path_at(im_i))
        roidb = self._load_mcg_maskdb(im_i)
        boxes = roidb['boxes']
        filter_keep = filter_small_boxes(boxes, min_size=16)
        boxes = boxes[filter_keep, :]
        masks = roidb['masks'][filter_keep, :, :]
        assert boxes.shape[0] == masks.shape[0]
        size = cfg.TEST.CFM_INPUT_MASK_SIZE
        # cv2.resize(mask, (size, size)) of every proposal mask (:346-350), all masks in one pass (as channels of one image)
        masks = resize_to(masks.transpose(1, 2, 0).astype(np.float32), size, size).transpose(2, 0, 1).astype(np.float64) \
            if masks.shape[0] else np.zeros((0, size, size))
        if cfg.TEST.USE_TOP_K_MCG:
            num_keep = min(boxes.shape[0], cfg.TEST.USE_TOP_K_MCG)
            boxes, masks = boxes[:num_keep, :], masks[:num_keep, :, :]
        # multi-scale test: adjacent levels are grouped into one forward
        im_scale_factors = cfm_scale_factors(im.shape, cfg.TEST.SCALES)     # (the reference builds the whole pyramid for these)
        orig_boxes = boxes.copy()
        boxes = pred_rois_for_blob(boxes, im_scale_factors)
        group = cfg.TEST.GROUP_SCALE
        num_scale_iter = int(np.ceil(len(cfg.TEST.SCALES) / float(group)))
        lo_scale = 0
        res_boxes = np.zeros((0, 4), dtype=np.float32)
        res_masks = np.zeros((0, 1, cfg.MASK_SIZE, cfg.MASK_SIZE), dtype=np.float32)
        res_seg_scores = np.zeros((0, self.num_classes), dtype=np.float32)
        partial = getattr(self.net, 'supports_partial_forward', False)      # pycaffe's forward(start=...), see below
        for scale_iter in range(num_scale_iter):
            hi_scale = min(lo_scale + group, len(cfg.TEST.SCALES))
            inds_this_scale = np.where((boxes[:, 0] >= lo_scale) & (boxes[:, 0] < hi_scale))[0]
            if len(inds_this_scale) == 0:
                lo_scale += group
                continue
            max_rois = cfg.TEST.MAX_ROIS_GPU[scale_iter]
            boxes_this_scale = boxes[inds_this_scale, :]
            masks_this_scale = masks[inds_this_scale, :, :]
            # the batch index starts from the lowest level PRESENT (not from lo_scale), as in the reference (:381)
            boxes_this_scale[:, 0] -= min(boxes_this_scale[:, 0])
            if can_prep_on_device(self.net, im):
                data, _ = prep_im_for_blob_cfm_device(self.net, im, cfg.TEST.SCALES[lo_scale:hi_scale])
            else:
                data, _ = prep_im_for_blob_cfm(im, cfg.TEST.SCALES[lo_scale:hi_scale])
                data = data.astype(np.float32, copy=False)
            for test_iter, start in enumerate(range(0, boxes_this_scale.shape[0], max_rois)):
                end = min(start + max_rois, boxes_this_scale.shape[0])
                input_box = boxes_this_scale[start:end, :].astype(np.float32, copy=False)
                input_mask = masks_this_scale[start:end, :, :].reshape(end - start, 1, size, size).astype(np.float32, copy=False)
                input_mask = (input_mask >= cfg.BINARIZE_THRESH).astype(np.float32, copy=False)
                self.net.blobs['rois'].reshape(*input_box.shape)
                self.net.blobs['masks'].reshape(*input_mask.shape)
                if partial and test_iter > 0:
                    # same pyramid as the previous chunk: conv5_3 is still on the GPU, only the RoI heads are run again
                    blobs_out = self.net.forward(start='roi_pooling_conv5', rois=input_box, masks=input_mask)
                else:
                    self.net.blobs['data'].reshape(*data.shape)
                    blobs_out = self.net.forward(data=data, rois=input_box, masks=input_mask)
                output_mask = np.array(blobs_out['mask_prob'], dtype=np.float32)
                output_score = np.array(blobs_out['seg_cls_prob'], dtype=np.float32)
                res_masks = np.vstack((res_masks, output_mask.reshape(end - start, 1, cfg.MASK_SIZE, cfg.MASK_SIZE)))
                res_seg_scores = np.vstack((res_seg_scores, output_score))
            res_boxes = np.vstack((res_boxes, orig_boxes[inds_this_scale, :]))
            lo_scale += group
        return res_masks, res_boxes, res_seg_scores
