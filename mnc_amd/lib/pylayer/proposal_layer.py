"""`pylayer.proposal_layer.ProposalLayer` -- RPN outputs -> at most 300 RoIs (reference:
lib/pylayer/proposal_layer.py:27-175; TEST phase only, the backward pass is training-only).

bottom: rpn_cls_prob_reshape [1,2A,H,W], rpn_bbox_pred [1,4A,H,W], im_info [1,3]   top: rois [R,5] float32
Semantics kept from the reference: fg scores are channels A..2A-1; rows ordered (h, w, a); all anchors are decoded,
none is dropped for crossing the border; clip to im_info[:2]; keep sides >= RPN_MIN_SIZE*scale; descending
argsort -> top 6000 -> NMS(0.7) -> top 300; batch index 0 is prepended; the top blob is reshaped inside forward."""
import numpy as np
import yaml

import caffe
from mnc_config import cfg
from nms.gpu_nms import gpu_nms
from nms.nms_wrapper import nms
from transform.anchors import generate_anchors, generate_shifted_anchors
from transform.bbox_transform import bbox_transform_inv, clip_boxes, filter_small_boxes


class ProposalLayer(caffe.Layer):
    def setup(self, bottom, top):
        params = yaml.safe_load(self.param_str_) or {}
        self._feat_stride = params["feat_stride"]
        self._anchors = generate_anchors()
        self._num_anchors = self._anchors.shape[0]
        self._grid_cache = {}
        top[0].reshape(1, 5)

    def reshape(self, bottom, top):
        pass  # shapes are data dependent; the top is reshaped in forward

    def _grid(self, height, width):
        key = (height, width)
        if key not in self._grid_cache:
            self._grid_cache[key] = generate_shifted_anchors(self._anchors, height, width, self._feat_stride)
        return self._grid_cache[key]

    def forward(self, bottom, top):
        assert bottom[0].data.shape[0] == 1, "Only single item batches are supported"
        phase = str(self.phase)
        pre_n, post_n = cfg[phase].RPN_PRE_NMS_TOP_N, cfg[phase].RPN_POST_NMS_TOP_N
        nms_thresh, min_size = cfg[phase].RPN_NMS_THRESH, cfg[phase].RPN_MIN_SIZE
        A = self._num_anchors
        probs = bottom[0].data
        deltas = bottom[1].data
        im_info = bottom[2].data[0, :]
        height, width = probs.shape[-2:]

        scores = probs[:, A:, :, :].transpose((0, 2, 3, 1)).reshape((-1, 1))
        deltas = deltas.transpose((0, 2, 3, 1)).reshape((-1, 4))
        proposals = bbox_transform_inv(self._grid(height, width), deltas)
        proposals, _ = clip_boxes(proposals, im_info[:2])
        big = filter_small_boxes(proposals, min_size * im_info[2])
        proposals, scores = proposals[big, :], scores[big]

        order = np.argsort(-scores.ravel(), kind="stable")     # score descending, ties by ascending index
        if pre_n > 0:
            order = order[:pre_n]
        proposals, scores = proposals[order, :], scores[order]

        dets = np.hstack((proposals, scores))
        if cfg.USE_GPU_NMS and dets.shape[0] > 0 and post_n > 0:
            # only keep[:post_n] is used below, so the device scan may stop there (identical prefix)
            keep = gpu_nms(dets, nms_thresh, device_id=cfg.GPU_ID, max_keep=post_n)
        else:
            keep = nms(dets, nms_thresh)
        if post_n > 0:
            keep = keep[:post_n]
        proposals = proposals[keep, :]
        self._proposal_index = keep

        rois = np.hstack((np.zeros((proposals.shape[0], 1), dtype=np.float32),
                          proposals.astype(np.float32, copy=False)))
        top[0].reshape(*rois.shape)
        top[0].data[...] = rois

    def backward(self, top, propagate_down, bottom):
        raise NotImplementedError("ProposalLayer.backward is training-only and outside this inference path")
