"""TEST DOUBLE (CPU tests only): an emulation of libmnc_hip.so's entry points on host memory, used to exercise the HOST
logic of mnc_amd.engine / mnc_amd.lib (graph planning, fusions, layouts, Concat views, Python-layer plumbing) where no
GPU exists.  "Device pointers" are addresses of numpy buffers, so pointer arithmetic done by the engine works as on
the device.  The arithmetic comes from torch / the oracle -- this file says nothing about the kernels; the -m gpu tests
do.  It is never importable from mnc_amd/ and is installed only by the `fake_gpu` fixture below."""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from oracle import native

_bufs = {}


def _f(ptr, shape):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, np.float32)
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr))).reshape(shape)


def _i(ptr, shape):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((ctypes.c_int * n).from_address(int(ptr))).reshape(shape)


def _write_ptr(addr, value):
    ctypes.c_void_p.from_address(int(addr)).value = value


def _c8(x):                      # [C,H,W] -> [C/8,H,W,8]
    C, H, W = x.shape
    return x.reshape(C // 8, 8, H, W).transpose(0, 2, 3, 1)


def _unc8(x):                    # [C/8,H,W,8] -> [C,H,W]
    CB, H, W, _ = x.shape
    return x.transpose(0, 3, 1, 2).reshape(CB * 8, H, W)


def _act(y, act):
    return F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _h16(ptr, shape):
    """A packed fp16 tensor at a "device" address."""
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(int(ptr))).view(np.float16).reshape(shape)


def _bf16_bits(a):
    """float32 -> bf16 bit patterns (nearest even), uint16."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _rd_c8(ptr, C, H, W, packed):
    """[C,H,W] float32 from a c8 tensor that is fp32 (packed 0 / False), packed fp16 (1 / True), or the test double's stand-ins for
    the split-bf16 form ('x3': the fp32 values as they are, 4 bytes a value) and plain bf16 ('bf16': bf16 bit patterns)."""
    shp = (C // 8, H, W, 8)
    if packed == "bf16":
        n = int(np.prod(shp))
        u = np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(int(ptr))).astype(np.uint32) << 16
        return _unc8(u.view(np.float32).reshape(shp))
    x = _h16(ptr, shp).astype(np.float32) if (packed and packed != "x3") else _f(ptr, shp)
    return _unc8(x)


def _wr_c8(ptr, y, packed):
    """[C,H,W] float32 -> c8 tensor, rounded to fp16 (nearest even) when packed (see _rd_c8 for 'x3' / 'bf16')."""
    C, H, W = y.shape
    if packed == "bf16":
        n = C * H * W
        np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(int(ptr)))[...] = _bf16_bits(_c8(y)).reshape(-1)
    elif packed and packed != "x3":
        _h16(ptr, (C // 8, H, W, 8))[...] = _c8(y).astype(np.float16)
    else:
        _f(ptr, (C // 8, H, W, 8))[...] = _c8(y)


class Fake(object):
    def mnc_device_count(self, addr):
        ctypes.c_int.from_address(int(addr)).value = 1

    def mnc_ctx_create(self, addr, dev):
        _write_ptr(addr, 0xC0FFEE)

    def mnc_ctx_destroy(self, h):
        pass

    def mnc_ctx_set_layer_conventions(self, h, addr):
        """The fake has one context: the conventions go straight to the oracle's switches (oracle.native.conventions)."""
        from mnc_amd.native_net import LayerConventions
        c = LayerConventions.from_address(int(addr)) if addr else LayerConventions.make()
        native._conv.clear()
        native._conv.update(c.as_dict())

    def mnc_ctx_sync(self, h):
        pass

    def mnc_dev_alloc(self, h, nbytes, addr):
        buf = np.zeros(int(nbytes) + 64, np.uint8)
        p = buf.ctypes.data
        _bufs[p] = buf
        _write_ptr(addr, p)

    def mnc_dev_free(self, h, p):
        _bufs.pop(int(p), None)

    def mnc_h2d(self, h, dst, src, n):
        ctypes.memmove(int(dst), int(src), int(n))

    mnc_d2h = mnc_h2d
    mnc_d2d = mnc_h2d
    mnc_h2d_async = mnc_h2d
    mnc_d2h_async = mnc_h2d

    def mnc_prof_enable(self, h, e):
        pass

    def mnc_prof_reset(self, h):
        pass

    def mnc_prof_count(self, h, addr):
        ctypes.c_int.from_address(int(addr)).value = 0

    # ---- layouts / packing ----
    def mnc_nchw_to_c8(self, h, src, dst, C, H, W):
        _f(dst, (C // 8, H, W, 8))[...] = _c8(_f(src, (C, H, W)))

    def mnc_c8_to_nchw(self, h, src, dst, C, H, W):
        _f(dst, (C, H, W))[...] = _unc8(_f(src, (C // 8, H, W, 8)))

    def mnc_rchw_to_rhwc(self, h, src, dst, R, C, PH, PW):
        _f(dst, (R, PH, PW, C))[...] = _f(src, (R, C, PH, PW)).transpose(0, 2, 3, 1)

    def mnc_rhwc_to_rchw(self, h, src, dst, R, C, PH, PW):
        _f(dst, (R, C, PH, PW))[...] = _f(src, (R, PH, PW, C)).transpose(0, 3, 1, 2)

    def mnc_pack_conv3x3_weights(self, h, src, dst, Cout, Cin):
        w = _f(src, (Cout, Cin, 3, 3))
        out = _f(dst, (Cin // 8, Cout, 76))
        out[...] = 0
        for cb in range(Cin // 8):
            for tap in range(9):
                out[cb, :, tap * 8:tap * 8 + 8] = w[:, cb * 8:cb * 8 + 8, tap // 3, tap % 3]

    def mnc_pack_fc_weights(self, h, src, dst, N, C, PH, PW):
        _f(dst, (N, PH * PW, C))[...] = _f(src, (N, C, PH * PW)).transpose(0, 2, 1)

    # ---- graph ops ----
    def mnc_conv3x3_c3(self, h, src, w, b, dst, H, W, Cout, relu):
        y = F.conv2d(_t(_f(src, (1, 3, H, W))), _t(_f(w, (Cout, 3, 3, 3))), _t(_f(b, (Cout,))), padding=1)
        _f(dst, (Cout // 8, H, W, 8))[...] = _c8(_act(y, relu)[0].numpy())

    def mnc_conv3x3(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        pk = _f(wpk, (Cin // 8, Cout, 76))
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        for cb in range(Cin // 8):
            for tap in range(9):
                w[:, cb * 8:cb * 8 + 8, tap // 3, tap % 3] = pk[cb, :, tap * 8:tap * 8 + 8]
        x = _unc8(_f(src, (Cin // 8, H, W, 8)))
        y = F.conv2d(_t(x)[None], _t(w), _t(_f(b, (Cout,))), padding=1)
        _f(dst, (Cout // 8, H, W, 8))[...] = _c8(_act(y, relu)[0].numpy())

    def mnc_pack_conv3x3_wino(self, h, src, dst, Cout, Cin):
        # test double: the packed buffer (Cin*Cout*17 floats) keeps the OIHW weights in its first Cout*Cin*9 floats; the real
        # transform G g G^T and the kernel are checked against torch on the GPU (tests/test_gpu_ops.py::test_conv3x3_winograd)
        _f(dst, (Cout * Cin * 9,))[...] = _f(src, (Cout * Cin * 9,))

    # F(4x4,3x3) (round 4): same test doubles, the real kernel is checked on the GPU (tests/test_gpu_ops.py::test_conv3x3_winograd_f4)
    # and emulated formula for formula on the CPU (tests/test_wino4_index_math.py)
    def mnc_pack_conv3x3_wino4(self, h, src, dst, Cout, Cin):
        return self.mnc_pack_conv3x3_wino(h, src, dst, Cout, Cin)

    def mnc_conv3x3_wino4(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        return self.mnc_conv3x3_wino(h, src, wpk, b, dst, H, W, Cin, Cout, relu)

    def mnc_conv3x3_wino4_pool(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        return self.mnc_conv3x3_wino_pool(h, src, wpk, b, dst, H, W, Cin, Cout, relu)

    def mnc_conv3x3_wino(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        w = _f(wpk, (Cout, Cin, 3, 3))
        x = _unc8(_f(src, (Cin // 8, H, W, 8)))
        y = F.conv2d(_t(x)[None], _t(w), _t(_f(b, (Cout,))), padding=1)
        _f(dst, (Cout // 8, H, W, 8))[...] = _c8(_act(y, relu)[0].numpy())

    def mnc_conv3x3_wino_pool(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        w = _f(wpk, (Cout, Cin, 3, 3))
        x = _unc8(_f(src, (Cin // 8, H, W, 8)))
        y = _act(F.conv2d(_t(x)[None], _t(w), _t(_f(b, (Cout,))), padding=1), relu)
        y = F.max_pool2d(y, 2, 2, ceil_mode=True)
        _f(dst, (Cout // 8, y.shape[2], y.shape[3], 8))[...] = _c8(y[0].numpy())

    def mnc_pack_conv3x3_bf16x3(self, h, src, dst, Cout, Cin):
        w = _f(src, (Cout, Cin // 8, 8, 9)).transpose(1, 0, 3, 2)                 # [cb][co][tap][8]
        hi = (np.ascontiguousarray(w).view(np.uint32) & 0xFFFF0000).view(np.float32)
        lo = (((w - hi).view(np.uint32) + 0x8000) & 0xFFFF0000).view(np.float32)
        out = np.ctypeslib.as_array(ctypes.cast(dst, ctypes.POINTER(ctypes.c_uint16)), (Cin // 8, Cout, 168))
        out[...] = 0
        body = out[:, :, :144].reshape(Cin // 8, Cout, 9, 2, 8)
        body[:, :, :, 0, :] = (hi.view(np.uint32) >> 16).astype(np.uint16)
        body[:, :, :, 1, :] = (lo.view(np.uint32) >> 16).astype(np.uint16)

    def mnc_conv3x3_bf16x3(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        pk = np.ctypeslib.as_array(ctypes.cast(wpk, ctypes.POINTER(ctypes.c_uint16)), (Cin // 8, Cout, 168))
        f = (pk[:, :, :144].astype(np.uint32) << 16).view(np.float32).reshape(Cin // 8, Cout, 9, 2, 8)
        wt = f[:, :, :, 0, :] + f[:, :, :, 1, :]                                    # [cb][co][tap][8]
        w = np.ascontiguousarray(wt.transpose(1, 0, 3, 2)).reshape(Cout, Cin, 3, 3)
        x = _unc8(_f(src, (Cin // 8, H, W, 8)))
        y = F.conv2d(_t(x)[None], _t(w), _t(_f(b, (Cout,))), padding=1)
        _f(dst, (Cout // 8, H, W, 8))[...] = _c8(_act(y, relu)[0].numpy())

    def mnc_pack_conv3x3_f16(self, h, src, dst, Cout, Cin):
        # test double: keep the fp16-rounded OIHW weights in the first Cout*Cin*9 floats of the packed buffer
        _f(dst, (Cout * Cin * 9,))[...] = _f(src, (Cout * Cin * 9,)).astype(np.float16).astype(np.float32)

    def mnc_conv3x3_f16(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu):
        x = _t(_unc8(_f(src, (Cin // 8, H, W, 8))).astype(np.float16).astype(np.float32))[None]
        y = F.conv2d(x, _t(_f(wpk, (Cout, Cin, 3, 3))), _t(_f(b, (Cout,))), padding=1)[0]
        if relu:
            y = F.relu(y)
        _f(dst, (Cout // 8, H, W, 8))[...] = _c8(y.numpy())

    def mnc_maxpool2_c8(self, h, src, dst, C, H, W):
        x = _unc8(_f(src, (C // 8, H, W, 8)))
        y = F.max_pool2d(_t(x)[None], 2, 2, ceil_mode=True)[0].numpy()
        _f(dst, (C // 8,) + y.shape[1:] + (8,))[...] = _c8(y)

    def mnc_conv1x1_to_nchw(self, h, src, w, b, dst, H, W, Cin, Cout):
        x = _unc8(_f(src, (Cin // 8, H, W, 8)))
        y = F.conv2d(_t(x)[None], _t(_f(w, (Cout, Cin, 1, 1))), _t(_f(b, (Cout,))))
        _f(dst, (Cout, H, W))[...] = y[0].numpy()

    def mnc_rpn_softmax(self, h, src, dst, A, H, W):
        s = _t(_f(src, (2, A * H * W)))
        _f(dst, (2, A * H * W))[...] = F.softmax(s, dim=0).numpy()

    def mnc_roi_warp(self, h, feat, C, H, W, rois, R, PH, PW, scale, pool2, dst):
        f = np.ascontiguousarray(_unc8(_f(feat, (C // 8, H, W, 8))))
        r = np.ascontiguousarray(_f(rois, (R, 5)))
        if pool2:
            out = native.maxpool2(native.roi_warp(f, r, 2 * PH, 2 * PW, scale))
        else:
            out = native.roi_warp(f, r, PH, PW, scale)
        _f(dst, (R, PH, PW, C))[...] = out.transpose(0, 2, 3, 1)

    def mnc_pack_conv_weights(self, h, src, dst, Cout, Cin, KH, KW):
        w = _f(src, (Cout, Cin // 8, 8, KH * KW))
        _f(dst, (KH * KW, Cin // 8, Cout, 8))[...] = w.transpose(3, 1, 0, 2)

    def mnc_conv2d(self, h, src, wpk, b, res, dst, H, W, Cin, Cout, KH, KW, stride, pad, relu):
        w = _f(wpk, (KH * KW, Cin // 8, Cout, 8)).transpose(2, 1, 3, 0).reshape(Cout, Cin, KH, KW)
        x = _t(_unc8(_f(src, (Cin // 8, H, W, 8))))[None]
        y = F.conv2d(x, _t(w), _t(_f(b, (Cout,))), stride=stride, padding=pad)[0]
        OH, OW = y.shape[1:]
        if res:
            y = y + _t(_unc8(_f(res, (Cout // 8, OH, OW, 8))))
        if relu:
            y = F.relu(y)
        _f(dst, (Cout // 8, OH, OW, 8))[...] = _c8(y.numpy())

    def mnc_pack_conv_weights_f16(self, h, src, dst, Cout, Cin, KH, KW):
        # test double: the fp16-rounded OIHW weights as floats at the start of the packed buffer (2 bytes/value were allocated
        # for ceil(Cin/32)*32 channels, so Cout*Cin*KH*KW floats fit only when padded -- keep a side table instead)
        self._f16_conv = getattr(self, "_f16_conv", {})
        self._f16_conv[int(dst)] = _f(src, (Cout, Cin, KH, KW)).astype(np.float16).astype(np.float32)

    def mnc_conv2d_f16(self, h, src, wpk, b, res, dst, H, W, Cin, Cout, KH, KW, stride, pad, relu):
        w = self._f16_conv[int(wpk)]
        x = _t(_unc8(_f(src, (Cin // 8, H, W, 8))).astype(np.float16).astype(np.float32))[None]
        y = F.conv2d(x, _t(w), _t(_f(b, (Cout,))), stride=stride, padding=pad)[0]
        OH, OW = y.shape[1:]
        if res:
            y = y + _t(_unc8(_f(res, (Cout // 8, OH, OW, 8))))
        if relu:
            y = F.relu(y)
        _f(dst, (Cout // 8, OH, OW, 8))[...] = _c8(y.numpy())

    def mnc_conv_stem_c3(self, h, src, w, b, dst, H, W, Cout, K, stride, pad, relu):
        y = F.conv2d(_t(_f(src, (1, 3, H, W))), _t(_f(w, (Cout, 3, K, K))), _t(_f(b, (Cout,))), stride=stride, padding=pad)[0]
        if relu:
            y = F.relu(y)
        _f(dst, (Cout // 8,) + tuple(y.shape[1:]) + (8,))[...] = _c8(y.numpy())

    def mnc_maxpool_c8(self, h, src, dst, C, H, W, K, stride, pad):
        y = F.max_pool2d(_t(_unc8(_f(src, (C // 8, H, W, 8))))[None], K, stride, pad, ceil_mode=True)[0].numpy()
        _f(dst, (C // 8,) + y.shape[1:] + (8,))[...] = _c8(y)

    # ---- 2-byte activation tensors of the "f16" mode and the 1x1 GEMM (csrc/conv1x1.hip) ----
    def mnc_act_pack(self, h, src, dst, n, f16):
        if f16 == 1:
            _h16(dst, (n,))[...] = _f(src, (n,)).astype(np.float16)
        elif f16 == 2:
            np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(int(dst)))[...] = _bf16_bits(_f(src, (n,)))
        else:                                               # split bf16 stand-in: the fp32 values
            _f(dst, (n,))[...] = _f(src, (n,)).copy()

    def mnc_act_unpack(self, h, src, dst, n, f16):
        if f16 == 1:
            _f(dst, (n,))[...] = _h16(src, (n,)).astype(np.float32)
        elif f16 == 2:
            u = np.ctypeslib.as_array((ctypes.c_uint16 * n).from_address(int(src))).astype(np.uint32) << 16
            _f(dst, (n,))[...] = u.view(np.float32)
        else:
            _f(dst, (n,))[...] = _f(src, (n,)).copy()

    def mnc_conv3x3_c3_fmt(self, h, src, w, b, dst, H, W, Cout, relu, out_fmt):
        y = F.conv2d(_t(_f(src, (1, 3, H, W))), _t(_f(w, (Cout, 3, 3, 3))), _t(_f(b, (Cout,))), padding=1)
        _wr_c8(dst, _act(y, relu)[0].numpy(), {0: False, 1: "x3", 2: True, 3: "bf16"}[out_fmt])

    def _conv_pk(self, kind, round_, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed):
        x = _t(round_(_rd_c8(src, Cin, H, W, kind if in_packed else False)))[None]
        if kind == "x3":                                    # mnc_pack_conv3x3_bf16x3's stand-in layout (above)
            pk = np.ctypeslib.as_array(ctypes.cast(wpk, ctypes.POINTER(ctypes.c_uint16)), (Cin // 8, Cout, 168))
            fw = (pk[:, :, :144].astype(np.uint32) << 16).view(np.float32).reshape(Cin // 8, Cout, 9, 2, 8)
            w = np.ascontiguousarray((fw[:, :, :, 0, :] + fw[:, :, :, 1, :]).transpose(1, 0, 3, 2)).reshape(Cout, Cin, 3, 3)
        else:
            w = _f(wpk, (Cout, Cin, 3, 3))
        y = F.conv2d(x, _t(w), _t(_f(b, (Cout,))), padding=1)[0]
        if relu:
            y = F.relu(y)
        _wr_c8(dst, y.numpy(), kind if out_packed else False)

    def mnc_conv3x3_bf16x3_pk(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed):
        self._conv_pk("x3", lambda a: a, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed)

    def mnc_conv3x3_bf16_pk(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed):
        rb = lambda a: (_bf16_bits(a).astype(np.uint32) << 16).view(np.float32).reshape(a.shape)
        self._conv_pk("bf16", rb, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed)

    def mnc_conv3x3_lowp_pool(self, h, mode, src, wpk, b, dst, H, W, Cin, Cout, relu):
        kind = {0: "x3", 1: True, 2: "bf16"}[mode]
        rnd = {0: (lambda a: a), 1: (lambda a: a.astype(np.float16).astype(np.float32)),
               2: (lambda a: (_bf16_bits(a).astype(np.uint32) << 16).view(np.float32).reshape(a.shape))}[mode]
        tmp = np.zeros((Cout // 8, H, W, 8), np.float32)
        self._conv_pk(kind, rnd, src, wpk, b, tmp.ctypes.data, H, W, Cin, Cout, relu, True, False)
        y = F.max_pool2d(_t(_unc8(tmp))[None], 2, 2, ceil_mode=True)[0].numpy()
        _wr_c8(dst, y, kind)

    def mnc_maxpool2_c8_bf16x3(self, h, src, dst, C, H, W):
        y = F.max_pool2d(_t(_rd_c8(src, C, H, W, "x3"))[None], 2, 2, ceil_mode=True)[0].numpy()
        _wr_c8(dst, y, "x3")

    def mnc_maxpool2_c8_bf16(self, h, src, dst, C, H, W):
        y = F.max_pool2d(_t(_rd_c8(src, C, H, W, "bf16"))[None], 2, 2, ceil_mode=True)[0].numpy()
        _wr_c8(dst, y, "bf16")

    def mnc_conv3x3_f16_pk(self, h, src, wpk, b, dst, H, W, Cin, Cout, relu, in_packed, out_packed):
        x = _t(_rd_c8(src, Cin, H, W, in_packed).astype(np.float16).astype(np.float32))[None]
        y = F.conv2d(x, _t(_f(wpk, (Cout, Cin, 3, 3))), _t(_f(b, (Cout,))), padding=1)[0]
        if relu:
            y = F.relu(y)
        _wr_c8(dst, y.numpy(), out_packed)

    def mnc_maxpool2_c8_f16(self, h, src, dst, C, H, W):
        y = F.max_pool2d(_t(_rd_c8(src, C, H, W, True))[None], 2, 2, ceil_mode=True)[0].numpy()
        _wr_c8(dst, y, True)

    def mnc_maxpool_c8_f16(self, h, src, dst, C, H, W, K, stride, pad):
        y = F.max_pool2d(_t(_rd_c8(src, C, H, W, True))[None], K, stride, pad, ceil_mode=True)[0].numpy()
        _wr_c8(dst, y, True)

    def mnc_conv_stem_c3_fmt(self, h, src, w, b, dst, H, W, Cout, K, stride, pad, relu, out_packed):
        y = F.conv2d(_t(_f(src, (1, 3, H, W))), _t(_f(w, (Cout, 3, K, K))), _t(_f(b, (Cout,))), stride=stride, padding=pad)[0]
        if relu:
            y = F.relu(y)
        _wr_c8(dst, y.numpy(), out_packed)

    def mnc_pack_conv_stem_f16(self, h, src, dst, Cout, K):
        self._stem16 = getattr(self, "_stem16", {})
        self._stem16[int(dst)] = _f(src, (Cout, 3, K, K)).astype(np.float16).astype(np.float32)

    def mnc_conv_stem_f16(self, h, src, wpk, b, dst, H, W, Cout, K, stride, pad, relu, out_packed):
        x = _t(_f(src, (1, 3, H, W)).astype(np.float16).astype(np.float32))
        y = F.conv2d(x, _t(self._stem16[int(wpk)]), _t(_f(b, (Cout,))), stride=stride, padding=pad)[0]
        if relu:
            y = F.relu(y)
        _wr_c8(dst, y.numpy(), out_packed)

    def mnc_pack_conv1x1(self, h, src, dst, Cout, Cin, f16):
        # test double: a side table keyed by the packed buffer's address (the fragment-order packing is checked on the GPU)
        w = _f(src, (Cout, Cin)).copy()
        self._c11 = getattr(self, "_c11", {})
        self._c11[int(dst)] = w.astype(np.float16).astype(np.float32) if f16 else w

    def _conv1x1(self, x, wpk, b, res, Cout, stride, relu):
        w = self._c11[int(wpk)]
        y = F.conv2d(_t(x)[None], _t(w[:, :, None, None]), _t(_f(b, (Cout,))), stride=stride)[0]
        if res is not None:
            y = y + _t(np.ascontiguousarray(res))
        return (F.relu(y) if relu else y).numpy()

    def mnc_conv1x1(self, h, src, wpk, b, res, dst, H, W, Cin, Cout, stride, relu):
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        r = _rd_c8(res, Cout, OH, OW, False) if res else None
        _wr_c8(dst, self._conv1x1(_rd_c8(src, Cin, H, W, False), wpk, b, r, Cout, stride, relu), False)

    def mnc_conv1x1_f16_pk(self, h, src, wpk, b, res, dst, H, W, Cin, Cout, stride, relu, res_packed, out_packed):
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        r = _rd_c8(res, Cout, OH, OW, res_packed) if res else None
        _wr_c8(dst, self._conv1x1(_rd_c8(src, Cin, H, W, True), wpk, b, r, Cout, stride, relu), out_packed)

    def mnc_add(self, h, a, b, dst, n, relu):
        y = _f(a, (n,)) + _f(b, (n,))
        _f(dst, (n,))[...] = np.maximum(y, 0) if relu else y

    def mnc_prep_image(self, h, im, H, W, means, x0, ax, OW, y0, ay, OH, out, PH, PW):
        img = np.ctypeslib.as_array((ctypes.c_ubyte * (H * W * 3)).from_address(int(im))).reshape(H, W, 3)
        m = np.ctypeslib.as_array((ctypes.c_double * 3).from_address(int(means)))
        f = (img.astype(np.float64) - m).astype(np.float32)
        xa, fx, ya, fy = _i(x0, (OW,)), _f(ax, (OW,)), _i(y0, (OH,)), _f(ay, (OH,))
        xb, yb = np.minimum(xa + 1, W - 1), np.minimum(ya + 1, H - 1)
        one = np.float32(1.0)
        rows = f[:, xa] * (one - fx)[None, :, None] + f[:, xb] * fx[None, :, None]
        o = rows[ya] * (one - fy)[:, None, None] + rows[yb] * fy[:, None, None]
        dst = _f(out, (3, PH, PW))
        dst[...] = 0
        dst[:, :OH, :OW] = o.transpose(2, 0, 1)

    def mnc_roi_pool(self, h, feat, N, C, H, W, rois, R, PH, PW, scale, dst):
        f = np.stack([_unc8(x) for x in _f(feat, (N, C // 8, H, W, 8))])
        out = native.roi_pool(np.ascontiguousarray(f), np.ascontiguousarray(_f(rois, (R, 5))), PH, PW, scale)
        _f(dst, (R, PH, PW, C))[...] = out.transpose(0, 2, 3, 1)

    def mnc_maxpool2_rhwc(self, h, src, dst, R, PH, PW, C):
        x = _f(src, (R, PH, PW, C)).transpose(0, 3, 1, 2)
        _f(dst, (R, PH // 2, PW // 2, C))[...] = native.maxpool2(x).transpose(0, 2, 3, 1)

    def mnc_mask_resize(self, h, src, dst, R, IH, IW, OH, OW):
        _f(dst, (R, 1, OH, OW))[...] = native.mask_resize(_f(src, (R, 1, IH, IW)), OH, OW)

    def mnc_mask_pool(self, h, feat, mask, dst, R, PH, PW, C, pool2):
        f = np.ascontiguousarray(_f(feat, (R, PH, PW, C)).transpose(0, 3, 1, 2))
        out = native.mask_pool(f, _f(mask, (R, 1, PH, PW)))
        if pool2:
            out = native.maxpool2(out)
        _f(dst, (R,) + out.shape[2:] + (C,))[...] = out.transpose(0, 2, 3, 1)

    def mnc_fc(self, h, a, w, b, dst, M, N, K, ldc, act):
        y = _act(F.linear(_t(_f(a, (M, K))), _t(_f(w, (N, K))), _t(_f(b, (N,)))), act).numpy()
        full = _f(dst, ((M - 1) * ldc + N,))
        for m in range(M):
            full[m * ldc: m * ldc + N] = y[m]

    def mnc_fc_pair(self, h, a0, w0, b0, dst0, a1, w1, b1, dst1, M, N, K, ldc, act):
        self.mnc_fc(h, a0, w0, b0, dst0, M, N, K, ldc, act)
        self.mnc_fc(h, a1, w1, b1, dst1, M, N, K, ldc, act)

    def mnc_pack_fc_bf16x3(self, h, src, dst, N, K):
        T, S = (N + 127) // 128, K // 32
        w = np.zeros((T * 128, K), np.float32)
        w[:N] = _f(src, (N, K))
        w = w.reshape(T, 128, S, 4, 8).transpose(0, 2, 1, 3, 4)                      # [tile][stage][row][group][8]
        hi = (np.ascontiguousarray(w).view(np.uint32) & 0xFFFF0000).view(np.float32)
        lo = (((w - hi).view(np.uint32) + 0x8000) & 0xFFFF0000).view(np.float32)
        out = np.ctypeslib.as_array(ctypes.cast(dst, ctypes.POINTER(ctypes.c_uint16)), (T, S, 128, 4, 2, 8))
        out[..., 0, :] = (hi.view(np.uint32) >> 16).astype(np.uint16)
        out[..., 1, :] = (lo.view(np.uint32) >> 16).astype(np.uint16)

    def mnc_fc_bf16x3(self, h, a, wpk, b, dst, M, N, K, ldc, act):
        T, S = (N + 127) // 128, K // 32
        pk = np.ctypeslib.as_array(ctypes.cast(wpk, ctypes.POINTER(ctypes.c_uint16)), (T, S, 128, 4, 2, 8))
        f = (pk.astype(np.uint32) << 16).view(np.float32)
        w = (f[..., 0, :] + f[..., 1, :]).transpose(0, 2, 1, 3, 4).reshape(T * 128, K)[:N]
        w = np.ascontiguousarray(w)
        self.mnc_fc(h, a, w.ctypes.data, b, dst, M, N, K, ldc, act)

    def mnc_pack_fc_f16(self, h, src, dst, N, K):
        w = _f(src, (N, K)).astype(np.float16)
        tiles = (N + 127) // 128
        pad = np.zeros((tiles * 128, K), np.float16)
        pad[:N] = w
        out = np.ctypeslib.as_array((ctypes.c_uint16 * (tiles * 128 * K)).from_address(int(dst)))
        out[...] = pad.reshape(tiles, 128, K // 64, 64).transpose(0, 2, 1, 3).reshape(-1).view(np.uint16)

    def mnc_fc_f16(self, h, a, wpk, b, dst, M, N, K, ldc, act):
        tiles = (N + 127) // 128
        raw = np.ctypeslib.as_array((ctypes.c_uint16 * (tiles * 128 * K)).from_address(int(wpk))).view(np.float16)
        w = raw.reshape(tiles, K // 64, 128, 64).transpose(0, 2, 1, 3).reshape(tiles * 128, K)[:N].astype(np.float32)
        x = _f(a, (M, K)).astype(np.float16).astype(np.float32)
        y = _act(F.linear(_t(x), _t(w), _t(_f(b, (N,)))), act).numpy()
        full = _f(dst, ((M - 1) * ldc + N,))
        for m in range(M):
            full[m * ldc:m * ldc + N] = y[m]

    # ---- stage-major 2-byte activation forms of the reduced-precision InnerProducts (Blob._sm) ----
    @staticmethod
    def _sm_write(dst, rows, fmt):
        """rows [M][K] float32 -> the test double's stand-in for the stage-major tensor: the rounded values as float32/float16 in
        row-major order (the real layout is checked on the GPU against mnc_fc_pack_act)."""
        M, K = rows.shape
        if fmt == 1:
            _h16(dst, (M, K))[...] = rows.astype(np.float16)
        elif fmt == 3:                                       # bf16 bit patterns (round 6)
            np.ctypeslib.as_array((ctypes.c_uint16 * (M * K)).from_address(int(dst)))[...] = _bf16_bits(rows).reshape(-1)
        else:
            _f(dst, (M, K))[...] = rows

    def mnc_roi_warp_sm(self, h, feat, C, H, W, rois, R, PH, PW, scale, pool2, dst, sm, fmt):
        if not dst:                                          # stage-major output only (round 6)
            assert sm and fmt
            self._tmp_warp = np.zeros((R * PH * PW * C,), np.float32)
            dst = self._tmp_warp.ctypes.data
        self.mnc_roi_warp(h, feat, C, H, W, rois, R, PH, PW, scale, pool2, dst)
        if sm and fmt:
            self._sm_write(sm, _f(dst, (R, PH * PW * C)), fmt)

    def mnc_maxpool2_rhwc_sm(self, h, src, dst, R, PH, PW, C, sm, fmt):
        self.mnc_maxpool2_rhwc(h, src, dst, R, PH, PW, C)
        if sm and fmt:
            self._sm_write(sm, _f(dst, (R, (PH // 2) * (PW // 2) * C)), fmt)

    def mnc_mask_pool_sm(self, h, feat, mask, dst, R, PH, PW, C, pool2, sm, fmt):
        self.mnc_mask_pool(h, feat, mask, dst, R, PH, PW, C, pool2)
        if sm and fmt:
            oh, ow = (PH // 2, PW // 2) if pool2 else (PH, PW)
            self._sm_write(sm, _f(dst, (R, oh * ow * C)), fmt)

    def mnc_box_mask_pool(self, h, feat, mask, box, mout, R, PH, PW, C, box_sm, mask_sm, fmt):
        if not box:                                          # stage-major outputs only (round 6)
            assert not mout and box_sm and mask_sm and fmt
            n = R * (PH // 2) * (PW // 2) * C
            self._tmp_pool = (np.zeros((n,), np.float32), np.zeros((n,), np.float32))
            box, mout = self._tmp_pool[0].ctypes.data, self._tmp_pool[1].ctypes.data
        self.mnc_maxpool2_rhwc_sm(h, feat, box, R, PH, PW, C, box_sm, fmt)
        self.mnc_mask_pool_sm(h, feat, mask, mout, R, PH, PW, C, 1, mask_sm, fmt)

    def mnc_roi_warp_sm_only_ok(self, h, C, pool2, ok):
        ctypes.c_int.from_address(int(ok)).value = 1

    @staticmethod
    def _sm_rows(sm, M, K, fmt):
        if fmt == 1:
            return _h16(sm, (M, K)).astype(np.float32)
        if fmt == 3:
            u = np.ctypeslib.as_array((ctypes.c_uint16 * (M * K)).from_address(int(sm))).astype(np.uint32) << 16
            return u.view(np.float32).reshape(M, K).copy()
        return _f(sm, (M, K)).copy()

    def mnc_fc_unpack_act(self, h, sm, dst, M, K, fmt):
        _f(dst, (M, K))[...] = self._sm_rows(sm, M, K, fmt)

    def mnc_box_mask_pool_ex(self, h, feat, feat_sm, feat_fmt, mask, box, mout, R, PH, PW, C, box_sm, mask_sm, fmt):
        if feat_sm and feat_fmt in (1, 2, 3) and box_sm and mask_sm and fmt:
            # the pooling reads the stage-major copy: the fp32 tensor rounded to fp16 / bf16 (what its producer wrote); the split-bf16
            # stand-in of this double keeps the fp32 values
            rows = self._sm_rows(feat_sm, R, PH * PW * C, feat_fmt)
            self._keep_feat = np.ascontiguousarray(rows)
            feat = self._keep_feat.ctypes.data
        self.mnc_box_mask_pool(h, feat, mask, box, mout, R, PH, PW, C, box_sm, mask_sm, fmt)

    def _fc_ex(self, fn, fmt16, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt):
        if sm:
            rows = _h16(sm, (mstride, K))[:M].astype(np.float32) if fmt16 else _f(sm, (mstride, K))[:M]
            a = np.ascontiguousarray(rows).ctypes.data
            self._keep = rows
        fn(h, a, wpk, b, dst, M, N, K, ldc, act)
        if osm and ofmt:
            full = _f(dst, ((M - 1) * ldc + N,))
            out = np.stack([full[m * ldc:m * ldc + N] for m in range(M)])
            self._sm_write(osm, out, ofmt)

    def mnc_fc_f16_ex(self, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt):
        self._fc_ex(self.mnc_fc_f16, True, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt)

    def mnc_fc_lowp_pair(self, h, mode, a0, sm0, a1, sm1, mstride, w0, w1, b0, b1, dst0, dst1, M, N, K, ldc, act, osm0, osm1, ofmt):
        fn = {0: self.mnc_fc_bf16x3_ex, 1: self.mnc_fc_f16_ex, 2: self.mnc_fc_bf16_ex}[mode]
        fn(h, a0, sm0, mstride, w0, b0, dst0, M, N, K, ldc, act, osm0, ofmt if osm0 else 0)
        fn(h, a1, sm1, mstride, w1, b1, dst1, M, N, K, ldc, act, osm1, ofmt if osm1 else 0)

    def mnc_fc_bf16_ex(self, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt):
        if sm:                                               # stage-major format 3: bf16 bit patterns (round 6)
            rows = self._sm_rows(sm, mstride, K, 3)[:M]
            self._keep = np.ascontiguousarray(rows)
            a = self._keep.ctypes.data
        self.mnc_fc_bf16(h, a, wpk, b, dst, M, N, K, ldc, act)
        if osm and ofmt:
            full = _f(dst, ((M - 1) * ldc + N,))
            self._sm_write(osm, np.stack([full[m * ldc:m * ldc + N] for m in range(M)]), ofmt)

    def mnc_fc_bf16x3_ex(self, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt):
        self._fc_ex(self.mnc_fc_bf16x3, False, h, a, sm, mstride, wpk, b, dst, M, N, K, ldc, act, osm, ofmt)

    def mnc_fc_f16_pre(self, h, sm, mstride, wpk, b, dst, M, N, K, ldc, act):
        a = np.ascontiguousarray(_h16(sm, (mstride, K))[:M].astype(np.float32))
        self.mnc_fc_f16(h, a.ctypes.data, wpk, b, dst, M, N, K, ldc, act)

    def mnc_fc_bf16x3_pre(self, h, sm, mstride, wpk, b, dst, M, N, K, ldc, act):
        a = np.ascontiguousarray(_f(sm, (mstride, K))[:M])
        self.mnc_fc_bf16x3(h, a.ctypes.data, wpk, b, dst, M, N, K, ldc, act)

    def mnc_softmax_rows(self, h, src, dst, M, N):
        _f(dst, (M, N))[...] = F.softmax(_t(_f(src, (M, N))), dim=1).numpy()

    def mnc_softmax_rows_ld(self, h, src, ld, dst, M, N):
        full = _f(src, ((M - 1) * ld + N,))
        rows = np.stack([full[m * ld: m * ld + N] for m in range(M)])
        _f(dst, (M, N))[...] = F.softmax(_t(rows), dim=1).numpy()

    def mnc_eltwise(self, h, src, dst, n, op):
        _f(dst, (n,))[...] = _act(_t(_f(src, (n,)).copy()), op).numpy()

    def mnc_copy2d(self, h, dst, dld, src, sld, rows, cols):
        d, s = _f(dst, ((rows - 1) * dld + cols,)), _f(src, ((rows - 1) * sld + cols,))
        for r in range(rows):
            d[r * dld:r * dld + cols] = s[r * sld:r * sld + cols]

    # ---- device-resident python layers ----
    def mnc_proposal(self, h, prob, bbox, A, H, W, anchors, stride, im_h, im_w, im_scale, pre, post, thr, min_size, rois,
                     num):
        from oracle import host as ohost
        im_info = np.array([[im_h, im_w, im_scale]], np.float32)
        props, scores = ohost.proposal_candidates(_f(prob, (1, 2 * A, H, W)), _f(bbox, (1, 4 * A, H, W)), im_info, stride)
        self._cand = (props.copy(), scores.ravel().copy())
        keep = native.nms_sorted(np.hstack((props, scores)), thr)[:post] if len(props) else []
        out = _f(rois, (post, 5))
        out[...] = 0
        out[:len(keep), 1:] = props[keep]
        self._nprop = len(keep)
        if num:
            ctypes.c_int.from_address(int(num)).value = len(keep)

    def mnc_proposal_count(self, h, num):
        ctypes.c_int.from_address(int(num)).value = self._nprop

    def mnc_proposal_count_ptr(self, h, addr):
        if getattr(self, "_nprop_buf", None) is None:
            self._nprop_buf = np.zeros(1, np.int32)
        self._nprop_buf[0] = self._nprop
        _write_ptr(addr, self._nprop_buf.ctypes.data)

    def mnc_host_alloc(self, h, nbytes, addr):
        self.mnc_dev_alloc(h, nbytes, addr)

    def mnc_host_free(self, h, p):
        self.mnc_dev_free(h, p)

    def mnc_proposal_candidates(self, h, boxes, scores, cap, n):
        b, s = self._cand
        ctypes.c_int.from_address(int(n)).value = len(b)
        if boxes:
            _f(boxes, b.shape)[...] = b
            _f(scores, s.shape)[...] = s

    def mnc_stage_bridge(self, h, rois, bbox, ldb, probs, ldp, R, K, im_h, im_w, out):
        from oracle import host as ohost
        fb, fp = _f(bbox, ((R - 1) * ldb + 4 * K,)), _f(probs, ((R - 1) * ldp + K,))
        bb = np.stack([fb[r * ldb: r * ldb + 4 * K] for r in range(R)])
        pp = np.stack([fp[r * ldp: r * ldp + K] for r in range(R)])
        _f(out, (R, 5))[...] = ohost.stage_bridge_forward_test(_f(rois, (R, 5)), bb, pp, np.array([[im_h, im_w, 1]], np.float32))

    # ---- b1 / b2 / b3 on host pointers ----
    def _nms(self, keep, num, boxes, n, dim, thr, max_keep):
        if n == 0:
            ctypes.c_int.from_address(int(num)).value = 0
            return
        k = native.nms_sorted(_f(boxes, (n, dim)), thr)
        if max_keep is not None and max_keep >= 0:
            k = k[:max_keep]
        _i(keep, (n,))[:len(k)] = k
        ctypes.c_int.from_address(int(num)).value = len(k)

    def mnc_nms(self, keep, num, boxes, n, dim, thr, dev):
        self._nms(keep, num, boxes, n, dim, thr, None)

    def mnc_nms_topk(self, keep, num, boxes, n, dim, thr, max_keep, dev):
        self._nms(keep, num, boxes, n, dim, thr, max_keep)

    def mnc_nms_batched(self, keep, num, boxes, n, dim, order, batch, thr, max_keep, dev):
        bx, od = _f(boxes, (n, dim)), _i(order, (batch, n))
        kp, nm = _i(keep, (batch, n)), _i(num, (batch,))
        for b in range(batch):
            k = native.nms_sorted(np.ascontiguousarray(bx[od[b]]), thr)
            if max_keep >= 0:
                k = k[:max_keep]
            kp[b, :len(k)] = k
            nm[b] = len(k)

    def mnc_mask_voting(self, boxes, masks, scores, order, n, K, S, max_per_image, nms_thr, iou_thr, H, W, omask, obox,
                        oscore, counts, rnum, dev):
        # test double: the oracle's step-by-step restatement of the reference (thresholds are its module constants)
        from oracle import host as ohost
        assert abs(nms_thr - ohost.MASK_MERGE_NMS_THRESH) < 1e-6 and abs(iou_thr - ohost.MASK_MERGE_IOU_THRESH) < 1e-6
        lm, lb = ohost.gpu_mask_voting(_f(masks, (n, 1, S, S)), _f(boxes, (n, 4)), _f(scores, (n, K)), K, max_per_image,
                                       W, H)
        R = sum(len(b) for b in lb)
        ctypes.c_int.from_address(int(rnum)).value = R
        _i(counts, (K - 1,))[...] = [len(b) for b in lb]
        if R:
            allb = np.concatenate(lb, 0)
            _i(obox, (R, 4))[...] = allb[:, :4].astype(np.int32)
            _f(oscore, (R,))[...] = allb[:, 4].astype(np.float32)
            _f(omask, (R, 1, S, S))[...] = np.concatenate(lm, 0)

    def mnc_mask_voting_dev(self, h, boxes, masks, scores, n, K, S, max_per_image, nms_thr, iou_thr, H, W, omask, obox,
                            oscore, counts, rnum):
        self.mnc_mask_voting(boxes, masks, scores, None, n, K, S, max_per_image, nms_thr, iou_thr, H, W, omask, obox, oscore,
                             counts, rnum, 0)

    def mnc_vote_instances(self, h, boxes, masks, scores, n, K, S, max_per_image, nms_thr, iou_thr, H, W, records, cap, counts):
        from oracle import host as ohost
        from mnc_amd.instances import records_from_lists
        lm, lb = ohost.gpu_mask_voting(_f(masks, (n, 1, S, S)), _f(boxes, (n, 4)), _f(scores, (n, K)), K, max_per_image, W, H)
        rec, total = records_from_lists(lm, lb, cap, S)
        _f(records, (cap, 6 + S * S))[...] = rec
        c = _i(counts, (K,))
        c[0] = total
        c[1:] = [len(b) for b in lb]

    # multi-GPU exchange: the double's "communicator" has one rank, whose all-gather is a copy
    def mnc_comm_unique_id(self, addr, n):
        ctypes.memset(int(addr), 7, 128)

    def mnc_comm_init(self, h, uid, nranks, rank):
        assert nranks == 1 and rank == 0

    def mnc_comm_info(self, h, nranks, rank, version):
        if version:
            ctypes.c_int.from_address(int(version)).value = 22705

    def mnc_gather_instances(self, h, send, recv, n):
        ctypes.memmove(int(recv), int(send), int(n) * 4)

    def mnc_comm_destroy(self, h):
        pass

    def mnc_dev_zero(self, h, p, n):
        ctypes.memset(int(p), 0, int(n))

    def mnc_detect_tail(self, h, rois1, R1, rois2, R2, scale, H, W, boxes):
        from oracle import host as ohost
        parts = []
        for ptr, R in ((rois1, R1), (rois2, R2)):
            if R:
                parts.append(ohost.clip_boxes(_f(ptr, (R, 5))[:, 1:5] / np.float32(scale), (H, W))[0])
        if parts:
            _f(boxes, (R1 + R2, 4))[...] = np.concatenate(parts, 0)

    def mnc_mv(self, boxes, masks, nb, inds, start, wts, nc, H, W, bd, S, R, omask, obox, dev):
        m, b = native.mv(_f(boxes, (nb, bd)), _f(masks, (nb, 1, S, S)), _i(inds, (nc,)) if nc else np.zeros(0, np.int32),
                         _i(start, (R,)), _f(wts, (nc,)) if nc else np.zeros(0, np.float32), H, W)
        _f(omask, (R, 1, S, S))[...] = m
        _i(obox, (R, 4))[...] = b

    def mnc_bbox_overlaps(self, boxes, n, query, k, out):
        b = np.ctypeslib.as_array((ctypes.c_double * (n * 4)).from_address(int(boxes))).reshape(n, 4)
        q = np.ctypeslib.as_array((ctypes.c_double * (k * 4)).from_address(int(query))).reshape(k, 4)
        np.ctypeslib.as_array((ctypes.c_double * (n * k)).from_address(int(out))).reshape(n, k)[...] = \
            native.bbox_overlaps(b, q)


def install(monkeypatch):
    from mnc_amd import _lib
    fake = Fake()

    fake.calls = {}                          # entry point -> number of calls (tests assert on the plan through it)

    def call(name, *args):
        fake.calls[name] = fake.calls.get(name, 0) + 1
        getattr(fake, name)(*args)
        return 0

    monkeypatch.setattr(_lib, "call", call)
    monkeypatch.setattr(_lib, "load", lambda: None)
    monkeypatch.setattr(_lib, "device_count", lambda: 1)
    return fake
