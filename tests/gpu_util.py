"""Helpers for the -m gpu tests: device buffers through the C ABI (no torch in the product path)."""
import ctypes

import numpy as np

from mnc_amd import _lib


class Dev(object):
    """One engine context + tracked device allocations."""

    def __init__(self, device_id=0):
        _lib.load()
        h = ctypes.c_void_p()
        _lib.call("mnc_ctx_create", ctypes.addressof(h), device_id)
        self.h = h.value
        self._ptrs = []

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        _lib.call("mnc_dev_alloc", self.h, int(nbytes), ctypes.addressof(p))
        self._ptrs.append(p.value)
        return p.value

    def put(self, arr, dtype=np.float32):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        p = self.alloc(max(arr.nbytes, 16))
        _lib.call("mnc_h2d", self.h, p, _lib.ptr(arr), arr.nbytes)
        return p

    def empty(self, shape, dtype=np.float32, fill=None):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.alloc(max(n, 16))
        if fill is not None:
            self.put_into(p, np.full(shape, fill, dtype=dtype))
        return p

    def put_into(self, p, arr):
        arr = np.ascontiguousarray(arr)
        _lib.call("mnc_h2d", self.h, p, _lib.ptr(arr), arr.nbytes)

    def get(self, p, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        _lib.call("mnc_d2h", self.h, _lib.ptr(out), p, out.nbytes)
        return out

    def call(self, name, *args):
        return _lib.call(name, self.h, *args)

    def sync(self):
        _lib.call("mnc_ctx_sync", self.h)

    def tune(self, name, value):
        """mnc_ctx_set_tuning: value None = the library's own choice again."""
        n = ctypes.c_char_p(name.encode())
        v = ctypes.c_char_p(str(value).encode()) if value is not None else None
        _lib.call("mnc_ctx_set_tuning", self.h, ctypes.cast(n, ctypes.c_void_p), ctypes.cast(v, ctypes.c_void_p) if v else None)

    def close(self):
        if self.h:
            for p in self._ptrs:
                _lib.call("mnc_dev_free", self.h, p)
            _lib.call("mnc_ctx_destroy", self.h)
            self.h = None


def to_c8(x):
    C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(C // 8, 8, H, W).transpose(0, 2, 3, 1))


def from_c8(x, C, H, W):
    return np.ascontiguousarray(x.reshape(C // 8, H, W, 8).transpose(0, 3, 1, 2).reshape(C, H, W))


def err(got, want):
    """(max abs error, max abs error relative to the reference's dynamic range)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    d = float(np.abs(got - want).max()) if got.size else 0.0
    scale = float(np.abs(want).max()) if want.size else 1.0
    return d, d / max(scale, 1e-30)
