"""A float32 array that lives in the engine's device memory and turns into numpy only when somebody looks at it.

`demo.im_detect` / `TesterWrapper._segmentation_forward` hand their results straight to `gpu_mask_voting`
(tools/demo.py:129-133, lib/caffeWrapper/TesterWrapper.py:166-196).  Returning DeviceArrays keeps that hand-over on the GPU
(mnc_detect_tail + mnc_mask_voting_dev) while every other consumer still gets what it expects: np.asarray(x), x[...],
x.shape, len(x), arithmetic -- all go through one cached device-to-host copy."""
import numpy as np

from . import _lib


class DeviceArray(object):
    __array_priority__ = 100.0

    def __init__(self, net, ptr, shape, keepalive=None, generation=None):
        """generation = (owner object, attribute name): the owner's counter is bumped whenever the device buffer behind this
        array is reused (the next image's detect_tail / prep_image); an array of an older generation that was never copied to
        the host refuses to hand out another image's data."""
        self._net = net
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(np.float32)
        self._keep = keepalive
        self._host = None
        self._gen = None if generation is None else (generation[0], generation[1], getattr(generation[0], generation[1]))

    def is_current(self):
        return self._gen is None or getattr(self._gen[0], self._gen[1]) == self._gen[2]

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)))

    def __len__(self):
        return self.shape[0]

    def numpy(self):
        if self._host is None:
            if not self.is_current():
                raise RuntimeError("this DeviceArray's buffer has been reused by a later image (copy results with np.asarray() "
                                   "before the next im_detect / prep_image if they must outlive it)")
            out = np.zeros(self.shape, np.float32)
            if out.size:
                _lib.call("mnc_d2h", self._net._ctx.h, _lib.ptr(out), self.ptr, out.nbytes)
            self._host = out
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def astype(self, dtype, copy=True):
        return self.numpy().astype(dtype, copy=copy)

    def copy(self):
        return self.numpy().copy()

    def __repr__(self):
        return "DeviceArray(shape=%r, float32, device)" % (self.shape,)
