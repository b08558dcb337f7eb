"""`NativeNet`: the whole-image C entry points (include/mnc_hip.h: mnc_net_* / mnc_forward_image; csrc/pipeline.hip) from Python.

The caffe-shaped `mnc_amd.engine.Net` executes any prototxt layer by layer from Python (the drop-in for `caffe.Net`); this class
drives the fixed-function form of ONE graph -- models/VGG16/mnc_5stage/test.prototxt, widths configurable -- where prep, forward,
the im_detect tail and gpu_mask_voting of an image are one C call (and, after the second image of a size, one HIP graph launch).
It exists for the bench and for tests; a non-Python host binds the same five functions."""
import ctypes

import numpy as np

from . import _lib
from .instances import split_records

MATH = {"fp32": 0, "bf16x3": 1, "f16": 2, "mixed": 3, "bf16": 4}
LAYERS = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1",
          "conv5_2", "conv5_3", "rpn_conv_3x3", "rpn_cls_score", "rpn_bbox_pred", "fc6_maskest", "mask_pred", "fc6", "fc7",
          "fc6_mask", "fc7_mask", "cls_score", "seg_cls_score", "bbox_pred"]


class LayerConventions(ctypes.Structure):
    """Mirror of `mnc_layer_conventions` (include/mnc_hip.h): the SPEC-CHOICEs of ROIWarping / MaskResize / MaskPooling."""
    _fields_ = [("warp_sample", ctypes.c_int), ("warp_round_edges", ctypes.c_int), ("warp_no_plus_one", ctypes.c_int),
                ("warp_oob", ctypes.c_int), ("resize_mode", ctypes.c_int), ("maskpool_binary", ctypes.c_int),
                ("maskpool_thresh", ctypes.c_float), ("inherit", ctypes.c_int)]

    @classmethod
    def make(cls, conv=None):
        """dict {field: value} (None / {} = the SPEC) -> struct with inherit = 0 (mnc_net_create APPLIES it); unknown names raise."""
        c = cls(0, 0, 0, 0, 0, 0, 0.4, 0)
        for k, v in (conv or {}).items():
            if k not in ("warp_sample", "warp_round_edges", "warp_no_plus_one", "warp_oob", "resize_mode", "maskpool_binary",
                         "maskpool_thresh"):
                raise KeyError("unknown layer convention %r" % k)
            setattr(c, k, float(v) if k == "maskpool_thresh" else int(v))
        return c

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "inherit"}


class NetConfig(ctypes.Structure):
    """Mirror of `mnc_net_config` (include/mnc_hip.h), field for field."""
    _fields_ = [("trunk_channels", ctypes.c_int * 5), ("rpn_channels", ctypes.c_int), ("num_anchors", ctypes.c_int),
                ("anchors", ctypes.c_float * 64), ("feat_stride", ctypes.c_int), ("pre_nms_topn", ctypes.c_int),
                ("post_nms_topn", ctypes.c_int), ("rpn_nms_thresh", ctypes.c_float), ("rpn_min_size", ctypes.c_float),
                ("mask_fc", ctypes.c_int), ("mask_size", ctypes.c_int), ("fc_dim", ctypes.c_int), ("num_classes", ctypes.c_int),
                ("roi_size", ctypes.c_int), ("spatial_scale", ctypes.c_float), ("target_size", ctypes.c_int),
                ("max_size", ctypes.c_int), ("pixel_means", ctypes.c_double * 3), ("max_per_image", ctypes.c_int),
                ("vote_nms_thresh", ctypes.c_float), ("vote_iou_thresh", ctypes.c_float), ("math", ctypes.c_int),
                ("use_graph", ctypes.c_int), ("winograd", ctypes.c_int), ("conventions", LayerConventions)]


def default_config():
    cfg = NetConfig()
    _lib.call("mnc_net_default_config", ctypes.addressof(cfg))
    return cfg


def config_from_weights(weights, math="fp32", use_graph=True, winograd=None, **overrides):
    """The reference's configuration (lib/mnc_config.py defaults) with the widths read off a weight dict {layer: [W, b]}."""
    cfg = default_config()
    stage_first = ["conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"]
    for i, name in enumerate(stage_first):
        cfg.trunk_channels[i] = int(weights[name][0].shape[0])
    cfg.rpn_channels = int(weights["rpn_conv_3x3"][0].shape[0])
    cfg.mask_fc = int(weights["fc6_maskest"][0].shape[0])
    cfg.fc_dim = int(weights["fc6"][0].shape[0])
    cfg.num_classes = int(weights["cls_score"][0].shape[0])
    cfg.math = MATH[math]
    cfg.use_graph = 1 if use_graph else 0
    from .engine import winograd_mode
    cfg.winograd = winograd_mode(winograd)             # 0 direct, 2 F(2x2,3x3), 4 F(4x4,3x3) (default)
    for k, v in overrides.items():
        if k == "layer_conventions":
            cfg.conventions = LayerConventions.make(v)
        elif k == "pixel_means":
            for i in range(3):
                cfg.pixel_means[i] = float(np.asarray(v).reshape(-1)[i])
        else:
            setattr(cfg, k, v)
    return cfg


class NativeNet(object):
    def __init__(self, weights, device_id=0, math="fp32", use_graph=True, winograd=None, **overrides):
        """weights: {layer: [W, b]} in Caffe layout (what mnc_amd.synth / caffemodel.load_weights return), or the path of a
        flat MNCW0001 file (caffemodel.save_flat) together with cfg=<NetConfig>, or ANOTHER NativeNet: this net then runs on that
        net's device weights and configuration (mnc_net_create_shared: own context, stream and buffers; nothing uploaded), which
        must stay open while this one is in use."""
        from .engine import _Ctx
        if isinstance(weights, NativeNet):
            parent = weights
            self._ctx = _Ctx(parent._ctx.device_id)
            self.cfg = parent.cfg
            self._parent = parent                       # (keeps the owner of the weights alive)
            h = ctypes.c_void_p()
            _lib.call("mnc_net_create_shared", self._ctx.h, parent.h, ctypes.addressof(h))
            self.h = h.value
            parent._sharers = getattr(parent, "_sharers", 0) + 1
            self.S = parent.S
            self.rec_dim = parent.rec_dim
            self.rows_cap = parent.rows_cap
            return
        cfg = overrides.pop("cfg", None)
        self._ctx = _Ctx(device_id)
        if cfg is None:
            cfg = config_from_weights(weights, math, use_graph, winograd, **overrides)
        self.cfg = cfg
        h = ctypes.c_void_p()
        _lib.call("mnc_net_create", self._ctx.h, ctypes.addressof(cfg), ctypes.addressof(h))
        self.h = h.value
        if isinstance(weights, str):
            path = weights.encode()
            _lib.call("mnc_net_load_file", self.h, ctypes.cast(ctypes.c_char_p(path), ctypes.c_void_p))
        else:
            for name in LAYERS:
                for idx in (0, 1):
                    a = np.ascontiguousarray(weights[name][idx], dtype=np.float32)
                    nm = name.encode()
                    _lib.call("mnc_net_set_param", self.h, ctypes.cast(ctypes.c_char_p(nm), ctypes.c_void_p), idx, _lib.ptr(a),
                              a.size)
        self.S = int(cfg.mask_size)
        self.rec_dim = 6 + self.S * self.S
        self.rows_cap = (int(cfg.num_classes) - 1) * int(cfg.max_per_image)

    def forward_image(self, im, record_cap=None):
        """uint8 BGR [H,W,3] -> (counts int32[num_classes], records float32[R, 6+S*S]): prep + forward + tail + voting, one call."""
        im = np.ascontiguousarray(im)
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise TypeError("forward_image takes a uint8 HxWx3 image (got %s %r)" % (im.dtype, im.shape))
        cap = self.rows_cap if record_cap is None else int(record_cap)
        rec = np.zeros((cap, self.rec_dim), np.float32)
        counts = np.zeros(int(self.cfg.num_classes), np.int32)
        _lib.call("mnc_forward_image", self.h, _lib.ptr(im), im.shape[0], im.shape[1], _lib.ptr(rec), cap, _lib.ptr(counts))
        return counts, rec[:min(int(counts[0]), cap)]

    def launch(self, im):
        """First half of forward_image: stage the image and enqueue the whole path on this net's stream; returns immediately."""
        im = np.ascontiguousarray(im)
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise TypeError("launch takes a uint8 HxWx3 image (got %s %r)" % (im.dtype, im.shape))
        _lib.call("mnc_forward_image_async", self.h, _lib.ptr(im), im.shape[0], im.shape[1], None, None)

    def fetch(self, record_cap=None):
        """Second half: wait for the launched image, -> (counts, records) as forward_image."""
        cap = self.rows_cap if record_cap is None else int(record_cap)
        rec = np.zeros((cap, self.rec_dim), np.float32)
        counts = np.zeros(int(self.cfg.num_classes), np.int32)
        _lib.call("mnc_net_fetch", self.h, _lib.ptr(rec), cap, _lib.ptr(counts))
        return counts, rec[:min(int(counts[0]), cap)]

    def detect(self, im):
        """-> (list_result_mask, list_result_box) as the reference's gpu_mask_voting returns them."""
        counts, rec = self.forward_image(im)
        return split_records(rec, counts[1:], self.S)

    def block(self):
        """The device-resident instance records of the last image as what InstanceGatherer.gather_block sends."""
        import types
        p, dims, nd = ctypes.c_void_p(), (ctypes.c_int * 4)(), ctypes.c_int(0)
        _lib.call("mnc_net_blob", self.h, ctypes.cast(ctypes.c_char_p(b"records"), ctypes.c_void_p), ctypes.addressof(p),
                  ctypes.addressof(dims), ctypes.addressof(nd))
        # (csrc/pipeline.hip: mnc_net::outblk = [counts 256 B | proposal count 256 B | records]; counts[0] = rows of this image)
        return types.SimpleNamespace(records_ptr=p.value, counts_ptr=p.value - 512, gather_rows=int(self.cfg.max_per_image),
                                     rec_dim=self.rec_dim, rows_cap=self.rows_cap)

    def blob(self, name):
        """(host copy, shape) of an intermediate blob of the last image in the engine's device layout (tests)."""
        p, dims, nd = ctypes.c_void_p(), (ctypes.c_int * 4)(), ctypes.c_int(0)
        nm = name.encode()
        _lib.call("mnc_net_blob", self.h, ctypes.cast(ctypes.c_char_p(nm), ctypes.c_void_p), ctypes.addressof(p),
                  ctypes.addressof(dims), ctypes.addressof(nd))
        shape = tuple(int(dims[i]) for i in range(nd.value))
        out = np.zeros(shape, np.float32)
        if out.size:
            _lib.call("mnc_d2h", self._ctx.h, _lib.ptr(out), p.value, out.nbytes)
        return out

    def profile(self, enable=True):
        _lib.call("mnc_prof_enable", self._ctx.h, int(enable))
        _lib.call("mnc_prof_reset", self._ctx.h)

    def profile_enable(self, level):
        """Switch event recording on (1 every launch, 2 MFMA launches) / off without touching the records collected so far
        (no synchronisation; profile() starts a fresh list, profile_records() drains it)."""
        _lib.call("mnc_prof_enable", self._ctx.h, int(level))

    def profile_records(self):
        n = ctypes.c_int(0)
        _lib.call("mnc_prof_count", self._ctx.h, ctypes.addressof(n))
        out = []
        name = ctypes.create_string_buffer(64)
        ms, fl, by = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_double(0)
        for i in range(n.value):
            _lib.call("mnc_prof_get", self._ctx.h, i, ctypes.addressof(name), 64, ctypes.addressof(ms), ctypes.addressof(fl),
                      ctypes.addressof(by))
            out.append((name.value.decode(), float(ms.value), float(fl.value), float(by.value)))
        _lib.call("mnc_prof_reset", self._ctx.h)
        return out

    def sync(self):
        _lib.call("mnc_ctx_sync", self._ctx.h)

    def arena_generation(self):
        """How often the context re-allocated one of its internal device arenas (a captured graph is dropped when it moves)."""
        g = ctypes.c_ulong(0)
        _lib.call("mnc_ctx_arena_generation", self._ctx.h, ctypes.addressof(g))
        return int(g.value)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_sharers", 0) > 0:
                raise RuntimeError("NativeNet.close: %d net(s) sharing this net's weights are still open" % self._sharers)
            _lib.call("mnc_net_destroy", self.h)
            self.h = None
            self._ctx.close()
            parent = getattr(self, "_parent", None)
            if parent is not None:
                parent._sharers -= 1
                self._parent = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImageStream(object):
    """Throughput form of NativeNet: `in_flight` nets (own context, stream and buffers each; ONE set of device weights, shared:
    mnc_net_create_shared) on ONE GPU, images
    submitted round-robin -- image k+1 is launched before image k is fetched, so the stretches of an image that occupy one or a
    few workgroups (proposal top-k, NMS scans, voting) run beside another image's convolutions.  Every image in flight has a
    hardware queue of its own (the library asks the runtime for 16: GPU_MAX_HW_QUEUES) and 1.2 GB of activation buffers; round 6,
    profiles/r06_streams.txt: 270 / 278 / 281 images/s at 600x1000 in fp32 with 4 / 8 / 12 in flight, 239 one at a time (the
    library's launch plans are made for this form -- least CU time per launch, MNC_PLAN=1 for the latency plans).  Results come back
    in submission order and equal NativeNet.forward_image's.

        stream = ImageStream(weights, in_flight=8)
        for counts, records in stream.map(images): ...
    """

    def __init__(self, weights, in_flight=8, **kwargs):
        if in_flight < 1:
            raise ValueError("in_flight must be >= 1")
        first = NativeNet(weights, **kwargs)
        self.nets = [first] + [NativeNet(first) for _ in range(int(in_flight) - 1)]
        self._pending = []                      # indices of the nets holding an unfetched image, oldest first
        self._next = 0

    def submit(self, im, record_cap=None):
        """Launch `im`; -> (counts, records) of the OLDEST image in flight when every net is busy, else None."""
        out = None
        if len(self._pending) == len(self.nets):
            out = self.nets[self._pending.pop(0)].fetch(record_cap)
        which = self._next
        self._next = (self._next + 1) % len(self.nets)
        self.nets[which].launch(im)
        self._pending.append(which)
        return out

    def drain(self, record_cap=None):
        """-> the results of the images still in flight, oldest first."""
        out = []
        while self._pending:
            out.append(self.nets[self._pending.pop(0)].fetch(record_cap))
        return out

    def map(self, images, record_cap=None):
        for im in images:
            r = self.submit(im, record_cap)
            if r is not None:
                yield r
        for r in self.drain(record_cap):
            yield r

    def close(self):
        for n in reversed(self.nets):                   # the sharers first, the owner of the weights last
            n.close()
        self.nets = []
