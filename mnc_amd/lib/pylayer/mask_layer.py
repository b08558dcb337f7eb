"""`pylayer.mask_layer.MaskLayer` -- TEST phase: sigmoid mask vector [R, S*S] -> mask_proposal [R,1,S,S]
(reference: lib/pylayer/mask_layer.py:22-29, 95-102).  The TRAIN branch (label assignment) is out of scope."""
import numpy as np

import caffe
from mnc_config import cfg


class MaskLayer(caffe.Layer):
    def setup(self, bottom, top):
        if str(self.phase) != "TEST":
            raise NotImplementedError("MaskLayer: only the TEST phase is implemented")
        top[0].reshape(1, 1, cfg.MASK_SIZE, cfg.MASK_SIZE)

    def reshape(self, bottom, top):
        pass

    def forward(self, bottom, top):
        pred = bottom[0].data
        out = pred.reshape((pred.shape[0], 1, cfg.MASK_SIZE, cfg.MASK_SIZE))
        top[0].reshape(*out.shape)
        top[0].data[...] = out.astype(np.float32, copy=False)

    def backward(self, top, propagate_down, bottom):
        raise NotImplementedError("MaskLayer.backward is training-only")
