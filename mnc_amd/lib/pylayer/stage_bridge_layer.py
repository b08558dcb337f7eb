"""`pylayer.stage_bridge_layer.StageBridgeLayer` -- TEST phase: refine each RoI with the box regressor of its
arg-max segmentation class (background column included) and clip (reference:
lib/pylayer/stage_bridge_layer.py:25-31, 52-55, 237-255).

bottom: rois [R,5], bbox_pred [R,4K], seg_cls_prob [R,K], im_info [1,3]      top: rois_ext [R,5] float32"""
import numpy as np
import yaml

import caffe
from transform.bbox_transform import bbox_transform_inv, clip_boxes


class StageBridgeLayer(caffe.Layer):
    def setup(self, bottom, top):
        yaml.safe_load(self.param_str_ or "")       # the TEST phase takes no parameters (empty param_str)
        if str(self.phase) != "TEST":
            raise NotImplementedError("StageBridgeLayer: only the TEST phase is implemented")
        top[0].reshape(1, 5)

    def reshape(self, bottom, top):
        pass

    def forward(self, bottom, top):
        rois = bottom[0].data
        decoded = bbox_transform_inv(rois[:, 1:5], bottom[1].data)          # [R, 4K]
        best = bottom[2].data.argmax(axis=1)
        cols = 4 * best[:, None] + np.arange(4)[None, :]
        picked = np.take_along_axis(decoded, cols, axis=1)
        out = np.zeros((rois.shape[0], 5), dtype=np.float64)                # float64 staging, stored as float32
        out[:, 1:5], _ = clip_boxes(picked.astype(np.float64), bottom[3].data[0, :2])
        top[0].reshape(*out.shape)
        top[0].data[...] = out.astype(np.float32, copy=False)

    def backward(self, top, propagate_down, bottom):
        raise NotImplementedError("StageBridgeLayer.backward is training-only")
