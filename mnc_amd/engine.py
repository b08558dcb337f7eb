"""Device-resident executor of MNC inference graphs (models/VGG16/mnc_5stage/test.prototxt) on MI355X.

`Net` reproduces the slice of pycaffe's `caffe.Net` that the reference's entry points and Python layers use
(tools/demo.py:70-90,126-129; lib/caffeWrapper/TesterWrapper.py:30-31,226-284):

    net = Net(prototxt, weights, caffe.TEST)        weights: .npz path | {layer: [W, b]} | {"layer/0": W, ...}
    net.blobs[name].reshape(*shape) / .data / .shape
    net.params[layer][i].data
    net.forward(data=..., im_info=...) -> {output blob: ndarray}
    net.name  (settable)

Every layer runs as a hand-written HIP kernel of libmnc_hip.so through the C ABI (include/mnc_hip.h); blobs stay on
the GPU in the engine's layouts (c8 feature maps, hwc per-RoI features) and are converted to Caffe's NCHW order only
when `.data` is read.  `type: 'Python'` layers are instantiated from their module/class exactly as Caffe does and
see host views of their bottoms/tops (this is where the reference also crosses device<->host).

There is no CPU execution path: constructing a Net without a working libmnc_hip.so + GPU raises."""
import ctypes
import importlib
import os
from collections import OrderedDict
from collections.abc import Mapping

import numpy as np

from . import _lib, install_paths, prototxt
from .devarray import DeviceArray

# fp32 mode: 3x3 convolutions by Winograd's minimal filtering on the fp32 matrix pipe.  MNC_CONV_WINOGRAD / Net(winograd=):
# 4 (default; True) = F(4x4,3x3), csrc/conv_wino4.hip (round 4); 2 (also "1") = F(2x2,3x3), csrc/conv_wino.hip; 0 (False) = direct
WINOGRAD_DEFAULT = "4"


def winograd_mode(value):
    """None / bool / 0 / 2 / 4 / "0" / "1" / "2" / "4" -> 0 (direct), 2 (F(2x2)) or 4 (F(4x4))."""
    import os
    if value is None:
        value = os.environ.get("MNC_CONV_WINOGRAD", WINOGRAD_DEFAULT)
    if value is True:
        return 4
    if value is False:
        return 0
    v = int(value)
    if v not in (0, 1, 2, 4):
        raise ValueError("winograd must be 0 (direct), 2 (F(2x2,3x3)) or 4 (F(4x4,3x3)), got %r" % (value,))
    return 2 if v == 1 else v

# bf16x3 mode: InnerProducts below this many flops stay on the fp32 kernel (its small-tile variant is as fast there)
_X3_MIN_FLOPS = 2.0e9

install_paths()

F32 = np.dtype(np.float32)


def _pool_out(n):
    """Caffe Pooling MAX 2x2/2 pad 0: ceil((n - 2) / 2) + 1."""
    return (n - 2 + 1) // 2 + 1


class _Ctx(object):
    def __init__(self, device_id):
        _lib.load()
        h = ctypes.c_void_p()
        _lib.call("mnc_ctx_create", ctypes.addressof(h), int(device_id))
        self.h = h.value
        self.device_id = int(device_id)
        self.allocs = 0                 # device allocations so far: a captured launch sequence is only replayed while this stands

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        _lib.call("mnc_dev_alloc", self.h, int(max(nbytes, 16)), ctypes.addressof(p))
        self.allocs += 1
        return p.value

    def free(self, p):
        if p and self.h:
            _lib.call("mnc_dev_free", self.h, p)

    def close(self):
        if self.h:
            _lib.call("mnc_ctx_destroy", self.h)
            self.h = None


class _DevBuf(object):
    """Growable device buffer."""

    def __init__(self, ctx):
        self.ctx, self.ptr, self.cap = ctx, 0, 0

    def ensure(self, nbytes):
        if nbytes > self.cap:
            if self.ptr:
                self.ctx.free(self.ptr)
            self.cap = int(nbytes * 1.25) + 256
            self.ptr = self.ctx.alloc(self.cap)
        return self.ptr

    def release(self):
        if self.ptr:
            self.ctx.free(self.ptr)
        self.ptr, self.cap = 0, 0


# Packed 2-byte-class activation layouts of the reduced-precision trunks (include/mnc_hip.h "Packed 2-byte activations"): the c8
# order in the mode's own operand form.  layout -> (bytes per value, mnc_act_pack format); conv math -> layout.
_PACKED = {"c8h": (2, 1), "c8x": (4, 0), "c8b": (2, 2)}
_PACKED_OF = {"f16": "c8h", "bf16x3": "c8x", "bf16": "c8b"}


class Blob(object):
    """A named tensor with Caffe's logical shape.  Device storage uses one of the engine layouts:
         'plain' row-major in Caffe order (optionally a column slice of a wider matrix: ld > shape[1])
         'c8'    [N][C/8][H][W][8]   for shape (N, C, H, W): N whole images back to back (N > 1 only in the CFM pyramid)
         'c8x' / 'c8b'  the same order in split bf16 ([hi x8 | lo x8] per pixel and 8 channels, 4 bytes a value) / plain bf16: the
                 "bf16x3" / "mixed" and "bf16" modes' trunk tensors (round 6), handled exactly like 'c8h'
         'c8h'   the same order in IEEE fp16 (the "f16" math mode's 2-byte activation tensors between trunk layers; the buffer
                 keeps its fp32 size, so the blob can be widened to 'c8' in place when a consumer or `.data` wants fp32)
         'rhwc'  [R][PH][PW][C]      for shape (R, C, PH, PW)
    Exactly one of (host, device) may be stale; `.data` makes the host copy current and hands it out."""

    def __init__(self, net, name, shape=(1,)):
        self._net = net
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self._host = None
        self._host_valid = False
        self._dev_valid = False
        self.layout = "plain"
        self._buf = _DevBuf(net._ctx)
        self._view = None           # (parent Blob, column offset) for Concat inputs written in place
        self.diff = None
        # second copy of a per-RoI tensor in the stage-major 2-byte form a reduced-precision InnerProduct multiplies from, written
        # by the tensor's producer (mnc_*_sm): {"fmt": 1 f16 | 2 bf16x3 | 3 bf16, "M": rows, "K": row length, "ptr": device address}
        self._sm = None
        self._smbuf = None
        # round 6: the producer wrote the stage-major form ONLY (mnc_roi_warp_sm / mnc_box_mask_pool_ex with null fp32 outputs); the
        # fp32 rows are materialised from it (mnc_fc_unpack_act) the first time a consumer or `.data` asks for them
        self._sm_only = False

    # ---- pycaffe surface ----
    @property
    def count(self):
        return int(np.prod(self.shape))

    @property
    def num(self):
        return self.shape[0]

    @property
    def channels(self):
        return self.shape[1] if len(self.shape) > 1 else 1

    def reshape(self, *dims):
        dims = tuple(int(d) for d in dims)
        if dims != self.shape:
            self.shape = dims
            self._host = None
            self._host_valid = False
            self._dev_valid = False
            self._sm_only = False

    def _materialize(self):
        """fp32 rows of a stage-major-only tensor (see __init__), in the layout its producer would have written."""
        sm = self._sm
        _lib.call("mnc_fc_unpack_act", self._net._ctx.h, sm["ptr"], self._buf.ensure(self.count * 4), sm["M"], sm["K"], sm["fmt"])
        self._dev_valid, self._sm_only = True, False

    @property
    def data(self):
        """Host ndarray in Caffe order.  Handing it out makes the host copy authoritative (it may be written)."""
        arr = self._host_read()
        self._dev_valid = False
        self._sm = None
        self._sm_only = False
        return arr

    # ---- engine side ----
    def _host_read(self):
        if not self._dev_valid and not self._host_valid and self._sm_only and self._sm is not None:
            self._materialize()
        if self._host is None or self._host.shape != self.shape:
            self._host = np.zeros(self.shape, dtype=F32)
            if not self._dev_valid:
                self._host_valid = True
        if not self._host_valid:
            if not self._dev_valid:
                raise RuntimeError("blob %r has no valid contents" % self.name)
            self._download()
            self._host_valid = True
        return self._host

    def set_host(self, arr):
        arr = np.ascontiguousarray(arr, dtype=F32)
        self.shape = arr.shape
        self._host = arr
        self._host_valid = True
        self._dev_valid = False
        self._sm = None
        self._sm_only = False

    def set_device(self, darr):
        """Adopt the contents of a DeviceArray of this net (plain layout): one device-to-device copy, no host round trip."""
        if darr._net is not self._net:
            raise ValueError("blob %r: the DeviceArray belongs to another net" % self.name)
        self.shape = tuple(darr.shape)
        self._host, self._host_valid = None, False
        if self.count:
            _lib.call("mnc_d2d", self._net._ctx.h, self._buf.ensure(self.count * 4), darr.ptr, self.count * 4)
        self.layout = "plain"
        self._dev_valid = True
        self._sm = None
        self._sm_only = False

    def _ld(self):
        if self._view is not None:
            return self._view[0].shape[1]
        return self.shape[1] if len(self.shape) > 1 else 1

    def dev_ptr(self):
        if self._view is not None:
            parent, off = self._view
            return parent._buf.ensure(parent.count * 4) + off * 4
        return self._buf.ensure(self.count * 4)

    def dev_out(self, layout):
        """Pointer for a kernel that is about to overwrite this blob in `layout`."""
        self.layout = layout
        self._dev_valid = True
        self._host_valid = False
        self._sm = None
        self._sm_only = False
        return self.dev_ptr()

    def dev_out_sm_only(self, layout):
        """The producer is about to write this blob in its stage-major form only (sm_out follows): no fp32 pointer."""
        self.layout = layout
        self._dev_valid = False
        self._host_valid = False
        self._sm = None
        self._sm_only = True
        return None

    def sm_out(self, fmt, M, K):
        """Device address for the producer's second output (stage-major 2-byte form, see __init__); call after dev_out."""
        if self._smbuf is None:
            self._smbuf = _DevBuf(self._net._ctx)
        ptr = self._smbuf.ensure(M * K * (4 if fmt == 2 else 2))
        self._sm = {"fmt": fmt, "M": M, "K": K, "ptr": ptr}
        return ptr

    def dev_in(self, layout):
        """Pointer to current contents in `layout`, uploading / converting as needed."""
        net = self._net
        if not self._dev_valid and self._sm_only and self._sm is not None:
            self._materialize()
        if not self._dev_valid:
            if not self._host_valid:
                raise RuntimeError("blob %r is read before it was produced" % self.name)
            assert self._view is None
            ptr = self._buf.ensure(self.count * 4)
            _lib.call("mnc_h2d", net._ctx.h, ptr, _lib.ptr(self._host), self.count * 4)
            self.layout = "plain"
            self._dev_valid = True
        if self.layout != layout:
            self._convert(layout)
        return self.dev_ptr()

    def _convert(self, layout):
        net = self._net
        tmp = net._tmp.ensure(self.count * 4)
        h = net._ctx.h
        src = self.dev_ptr()
        if self.layout in _PACKED or layout in _PACKED:
            # packed <-> fp32 in the c8 order is elementwise (batch and all); any other pairing goes through 'c8'
            if self.layout in _PACKED:
                _lib.call("mnc_act_unpack", h, src, tmp, self.count, _PACKED[self.layout][1])
                _lib.call("mnc_d2d", h, src, tmp, self.count * 4)
                self.layout = "c8"
                if layout != "c8":
                    self._convert(layout)
                return
            if self.layout != "c8":
                self._convert("c8")
            eb, fmt = _PACKED[layout]
            _lib.call("mnc_act_pack", h, src, tmp, self.count, fmt)
            _lib.call("mnc_d2d", h, src, tmp, self.count * eb)
            self.layout = layout
            return
        if self.layout == "plain" and layout == "c8":
            N, C, H, W = self.shape
            for n in range(N):
                _lib.call("mnc_nchw_to_c8", h, src + n * C * H * W * 4, tmp + n * C * H * W * 4, C, H, W)
        elif self.layout == "c8" and layout == "plain":
            N, C, H, W = self.shape
            for n in range(N):
                _lib.call("mnc_c8_to_nchw", h, src + n * C * H * W * 4, tmp + n * C * H * W * 4, C, H, W)
        elif self.layout == "plain" and layout == "rhwc":
            R, C, PH, PW = self.shape
            _lib.call("mnc_rchw_to_rhwc", h, src, tmp, R, C, PH, PW)
        elif self.layout == "rhwc" and layout == "plain":
            R, C, PH, PW = self.shape
            _lib.call("mnc_rhwc_to_rchw", h, src, tmp, R, C, PH, PW)
        else:
            raise RuntimeError("no conversion %s -> %s for blob %r" % (self.layout, layout, self.name))
        _lib.call("mnc_d2d", h, src, tmp, self.count * 4)
        self.layout = layout

    def _download(self):
        net = self._net
        h = net._ctx.h
        n = self.count
        if n == 0:
            return
        if self._view is not None:
            rows, cols = self.shape[0], self.shape[1]
            tmp = net._tmp.ensure(n * 4)
            _lib.call("mnc_copy2d", h, tmp, cols, self.dev_ptr(), self._ld(), rows, cols)
            _lib.call("mnc_d2h", h, _lib.ptr(self._host), tmp, n * 4)
            return
        if self.layout in _PACKED:
            self._convert("c8")          # widened in place: the values are unchanged, a later packed consumer re-packs them exactly
        if self.layout == "plain":
            _lib.call("mnc_d2h", h, _lib.ptr(self._host), self.dev_ptr(), n * 4)
            return
        tmp = net._tmp.ensure(n * 4)
        if self.layout == "c8":
            N, C, H, W = self.shape
            for i in range(N):
                _lib.call("mnc_c8_to_nchw", h, self.dev_ptr() + i * C * H * W * 4, tmp + i * C * H * W * 4, C, H, W)
        else:
            R, C, PH, PW = self.shape
            _lib.call("mnc_rhwc_to_rchw", h, self.dev_ptr(), tmp, R, C, PH, PW)
        _lib.call("mnc_d2h", h, _lib.ptr(self._host), tmp, n * 4)


class _ReadOnlyBlob(object):
    """What a Python layer gets as `bottom[i]`: reading .data does not invalidate the device copy."""

    def __init__(self, blob):
        self._b = blob

    @property
    def data(self):
        return self._b._host_read()

    @property
    def shape(self):
        return self._b.shape

    @property
    def diff(self):
        return None

    def reshape(self, *dims):
        self._b.reshape(*dims)


class _Param(object):
    def __init__(self, arr):
        self.data = arr
        self.shape = arr.shape


class _Layer(object):
    def __init__(self, msg):
        self.msg = msg
        self.name = msg.get1("name")
        self.type = msg.get1("type")
        self.bottoms = list(msg.all("bottom"))
        self.tops = list(msg.all("top"))
        self.param_names = [p.get1("name") for p in msg.all("param")]
        self.skip = False
        self.relu = False          # in-place ReLU folded into this layer
        self.act = 0               # 1 relu / 2 sigmoid folded into an InnerProduct
        self.fused_pool = False    # following MAX 2x2/2 folded into this ROIWarping / MaskPooling
        self.out_name = None       # blob written when a fusion redirects the output
        self.group = None          # [layers] of a merged sibling-InnerProduct GEMM (this layer is the leader)
        self.group_leader = None   # set on the followers of such a group
        self.bn = None             # BatchNorm / Scale layers folded into this Convolution's weights and bias
        self.scale = None
        self.residual = None       # blob added in this Convolution's epilogue (a following Eltwise SUM folded in)
        self.out_h = False         # "f16" math mode: this layer writes its top as a packed fp16 c8 tensor ('c8h')
        self.with_mask = None      # Pooling on a per-RoI tensor: the MaskPooling (+ its Pooling) of the same tensor done in the same pass
        self.pair = None           # InnerProduct: a later InnerProduct of the same shape computed in the same launch (mnc_fc_pair)
        self.pair_leader = None    # ... set on that later layer
        self.ip_prepare = None     # InnerProduct (fp32 kernel): () -> the arguments of its mnc_fc call, tops made ready
        self.pair_done = False     # per forward: the leader has computed this layer
        self.run = None


class _Outputs(Mapping):
    """What net.forward() returns: {output blob name: ndarray}, as pycaffe -- but an array is copied from the device when it
    is first looked at (tools/demo.py ignores the return value and reads net.blobs[...]; four eager copies per forward were
    four stream synchronisations)."""

    def __init__(self, net, names):
        self._net, self._names = net, list(names)

    def __getitem__(self, name):
        if name not in self._names:
            raise KeyError(name)
        return self._net.blobs[name]._host_read()

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __repr__(self):
        return "{%s}" % ", ".join("%r: <%s>" % (n, "x".join(map(str, self._net.blobs[n].shape))) for n in self._names)


class Net(object):
    supports_partial_forward = True      # forward(start=..., end=...) re-uses the blobs of the previous call

    def __init__(self, prototxt_path, weights, phase=1, device_id=None, fuse=None, native_pylayers=None, math=None,
                 winograd=None, layer_conventions=None):
        if device_id is None:
            try:
                import caffe
                device_id = caffe.get_device()
            except Exception:
                device_id = 0
        if _lib.device_count() <= device_id:
            raise RuntimeError("mnc_amd.engine.Net needs GPU %d (found %d device(s)); there is no CPU path"
                               % (device_id, _lib.device_count()))
        self.name = os.path.splitext(os.path.basename(str(prototxt_path)))[0]
        self.phase = "TEST" if phase in (1, "TEST") else "TRAIN"
        if self.phase != "TEST":
            raise NotImplementedError("only caffe.TEST graphs are executed")
        self._fuse = (os.environ.get("MNC_NO_FUSE", "0") != "1") if fuse is None else bool(fuse)
        # the three stock inference-time Python layers have device-resident equivalents (csrc/proposal.hip); any other
        # `type: 'Python'` layer -- or all of them with native_pylayers=False / MNC_NATIVE_PYLAYERS=0 -- runs as Python
        self._native_py = ((os.environ.get("MNC_NATIVE_PYLAYERS", "1") != "0") if native_pylayers is None
                           else bool(native_pylayers))
        # arithmetic of the dense contractions: "fp32" = fp32 MFMA throughout (the default and the parity reference);
        # "bf16x3" = the large InnerProducts and 3x3 convolutions run on the bf16 matrix pipe with every operand split
        # into hi + lo bf16 and three products per term (fp32-class accuracy, see csrc/gemm_x3.hip); "f16" = the same
        # layers in plain fp16 (operands rounded to fp16, fp32 accumulate: one product per term, ~3e-4 relative error per
        # layer -- the reduced-precision mode BASELINE configs[4] names); math= / MNC_MATH
        self.math = (os.environ.get("MNC_MATH", "fp32") if math is None else math).lower()
        if self.math not in ("fp32", "bf16x3", "f16", "mixed", "bf16"):
            raise ValueError("math must be 'fp32', 'bf16x3', 'f16', 'mixed' or 'bf16', got %r" % self.math)
        # "bf16" (round 4): BASELINE configs[2] as written -- the 3x3 convolutions and the large InnerProducts with both operands
        # rounded to bf16, ONE product per term (mnc_conv3x3_bf16 / mnc_fc_bf16), fp32 tensors between the layers.  Measured, not
        # recommended: 8 mantissa bits put the network far outside the 1e-3 bar (tests/test_gpu_parity8.py records it).
        # "mixed" (round 4): the convolutions (trunk, RPN head) in bf16x3 -- fp32-class -- and the large InnerProducts in fp16: the
        # reduced-precision mode that keeps the 1e-3 bar (13 fp16 trunk layers in a row leave conv5_3 at 1e-3 of its range and the
        # RPN probabilities at 3.6e-3; behind a bf16x3 trunk the two fp16 InnerProducts of a head branch cost ~3e-4)
        self.conv_math = "bf16x3" if self.math == "mixed" else self.math
        self.fc_math = "f16" if self.math == "mixed" else self.math
        self._wino = winograd_mode(winograd)
        self._winograd = self._wino != 0
        # MNC_SPECULATE_ROIS=0: read the ProposalLayer's RoI count back before the heads are launched (one stream sync in the
        # middle of forward) instead of running the heads on RPN_POST_NMS_TOP_N rows and checking the count at the end
        self._speculate = os.environ.get("MNC_SPECULATE_ROIS", "1") != "0"
        self._speculated = None
        self._running = 0
        self._ctx = _Ctx(device_id)
        # ROIWarping / MaskResize / MaskPooling conventions (include/mnc_hip.h: mnc_layer_conventions; oracle/SPEC.md section 6):
        # layer_conventions= {field: value}, else cfg.LAYER_CONVENTIONS (lib/mnc_config.py); {} = the SPEC.  Set on the context:
        # the RoI kernels' launchers read them there.
        if layer_conventions is None:
            try:
                from mnc_config import cfg as _cfg
                layer_conventions = dict(_cfg.get("LAYER_CONVENTIONS", {}) or {})
            except ImportError:
                layer_conventions = {}
        from .native_net import LayerConventions
        self.layer_conventions = LayerConventions.make(layer_conventions)
        _lib.call("mnc_ctx_set_layer_conventions", self._ctx.h, ctypes.addressof(self.layer_conventions))
        self._tmp = _DevBuf(self._ctx)
        self._net_msg = prototxt.parse_file(prototxt_path)
        self._layers = [_Layer(m) for m in self._net_msg.all("layer")]
        self.blobs = OrderedDict()
        self.params = OrderedDict()
        self._dev_params = {}
        self._py = {}
        self.inputs = list(self._net_msg.all("input"))
        for name, shp in zip(self.inputs, self._net_msg.all("input_shape")):
            self.blobs[name] = Blob(self, name, shp.all("dim"))
        for L in self._layers:
            for t in L.tops:
                if t not in self.blobs:
                    self.blobs[t] = Blob(self, t)
        consumed = set(b for L in self._layers for b in L.bottoms)
        self.outputs = [n for n in self.blobs if n not in consumed and n not in self.inputs]
        self._consumers = {}
        for i, L in enumerate(self._layers):
            for b in L.bottoms:
                if b not in L.tops:           # in-place layers do not count as separate consumers
                    self._consumers.setdefault(b, []).append(i)
        self._plan_fusions()
        self._warn_unpinned_layers(weights)
        self._host_weights = self._read_weights(weights)
        self._plan_formats()
        self._bind_layers()
        _lib.call("mnc_ctx_sync", self._ctx.h)

    # ------------------------------------------------------------------------------------------------ weights
    UNPINNED_LAYERS = ("ROIWarping", "MaskResize", "MaskPooling", "ROIPooling")

    def _warn_unpinned_layers(self, weights):
        """Trained Caffe weights + layers whose source is in the absent caffe-mnc submodule: say LOUDLY that their sampling
        conventions (oracle/SPEC.md: un-rounded RoI edges, +1 widths, top-left-aligned taps, zero taps outside the map) are this
        project's reading of the paper, not caffe-mnc's code -- masks and scores may differ from upstream until a golden vector from
        real caffe-mnc outputs pins them.  MNC_ACCEPT_UNPINNED_LAYERS=1 acknowledges and silences."""
        if isinstance(weights, dict) or not str(weights).endswith((".caffemodel", ".h5")):
            return
        kinds = sorted({L.type for L in self._layers if L.type in self.UNPINNED_LAYERS})
        if kinds and os.environ.get("MNC_ACCEPT_UNPINNED_LAYERS", "0") != "1":
            import warnings
            warnings.warn("PARITY UNPINNED: %s is loaded into a graph with %s layers.  Their semantics follow oracle/SPEC.md "
                          "(the caffe-mnc sources are not available to this project); outputs of trained weights may differ "
                          "from upstream caffe-mnc.  Set MNC_ACCEPT_UNPINNED_LAYERS=1 to acknowledge."
                          % (os.path.basename(str(weights)), ", ".join(kinds)), UserWarning, stacklevel=3)

    @staticmethod
    def _read_weights(weights):
        """dict {"<layer>": [W, b]} / {"<layer>/<i>": array}, or a path to .npz / .caffemodel / .caffemodel.h5 (the two
        containers the reference loads, tools/demo.py:46-48; read by mnc_amd/caffemodel.py without protobuf / h5py)."""
        if isinstance(weights, dict):
            src = weights
        else:
            from . import caffemodel
            src = caffemodel.load_weights(weights)
        out = {}
        for k, v in src.items():
            if isinstance(v, (list, tuple)):
                out[k] = [None if a is None else np.asarray(a, dtype=F32) for a in v]
            else:
                lname, idx = k.rsplit("/", 1)
                lst = out.setdefault(lname, [None, None])
                while len(lst) <= int(idx):
                    lst.append(None)
                lst[int(idx)] = np.asarray(v, dtype=F32)
        return out

    def _layer_weights(self, L):
        """[W, b] of a layer, following Caffe's sharing by `param { name }` (test.prototxt:514-515 <-> :829-834)."""
        if L.name in self._host_weights:
            return self._host_weights[L.name]
        for other in self._layers:
            if other is not L and L.param_names and all(L.param_names) and other.param_names == L.param_names \
                    and other.name in self._host_weights:
                return self._host_weights[other.name]
        raise KeyError("no weights for layer %r (param names %r)" % (L.name, L.param_names))

    def _upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=F32)
        p = self._ctx.alloc(arr.nbytes)
        _lib.call("mnc_h2d", self._ctx.h, p, _lib.ptr(arr), arr.nbytes)
        return p

    def _dev_param(self, key, builder):
        if key not in self._dev_params:
            self._dev_params[key] = builder()
        return self._dev_params[key]

    # ------------------------------------------------------------------------------------------------ planning
    def _sole_consumer(self, blob, typ):
        c = self._consumers.get(blob, [])
        if len(c) == 1 and self._layers[c[0]].type == typ and blob not in self.outputs:
            return self._layers[c[0]]
        return None

    def _plan_fusions(self):
        by_top = {}
        for L in self._layers:
            if L.type == "ReLU" and L.bottoms == L.tops:
                prod = by_top.get(L.bottoms[0])
                if prod is not None and prod.type in ("Convolution", "InnerProduct", "Eltwise"):
                    prod.relu = True
                    L.skip = True          # always folded: an in-place ReLU has no blob of its own
            elif L.type in ("BatchNorm", "Scale") and L.bottoms == L.tops:
                # conv -> BatchNorm(use_global_stats) -> Scale, all in place (the ResNet idiom): an affine map per output
                # channel, folded into the convolution's weights and bias when they are uploaded.  Always folded -- there
                # is no stand-alone kernel for them.
                prod = by_top.get(L.bottoms[0])
                ok = prod is not None and prod.type == "Convolution" and not prod.relu and prod.residual is None
                if ok and L.type == "BatchNorm" and prod.bn is None and prod.scale is None:
                    prod.bn, L.skip = L, True
                elif ok and L.type == "Scale" and prod.scale is None:
                    prod.scale, L.skip = L, True
                if L.skip:
                    continue               # by_top keeps naming the convolution as the producer of this blob
            for t in L.tops:
                by_top[t] = L
        if not self._fuse:
            return
        for L in self._layers:
            if L.type == "InnerProduct" and not L.relu:
                nxt = self._sole_consumer(L.tops[0], "Sigmoid")
                if nxt is not None:
                    L.act, L.out_name, nxt.skip = 2, nxt.tops[0], True
            # (round 6: in the reduced-precision modes too -- mnc_conv3x3_lowp_pool, packed tensors in and out, the convolution's own
            # bits through the pool's; what csrc/pipeline.hip's packed trunk launches)
            if (L.type == "Convolution" and ((self.conv_math == "fp32" and self._winograd) or
                                             (self.conv_math in _PACKED_OF and os.environ.get("MNC_F16_ACTS", "1") != "0"))
                    and L.bn is None and L.scale is None and L.residual is None and self._conv_kind(L) == "fast3x3"):
                # conv3x3 + ReLU + Pooling MAX 2x2/2 (conv1_2 / conv2_2 / conv3_3 / conv4_3 of the trunk): a 2x2 Winograd
                # output tile is a pooling window, the pool is applied in the convolution's epilogue (mnc_conv3x3_wino_pool)
                nxt = self._sole_consumer(L.tops[0], "Pooling")
                if nxt is not None and self._is_pool2(nxt):
                    L.fused_pool, L.out_name, nxt.skip = True, nxt.tops[0], True
            if L.type in ("ROIWarping", "MaskPooling"):
                nxt = self._sole_consumer(L.tops[0], "Pooling")
                if nxt is not None and self._is_pool2(nxt):
                    if L.type == "ROIWarping":
                        rp = L.msg.get1("roi_warping_param")
                        if rp.get1("pooled_h") % 2 or rp.get1("pooled_w") % 2:
                            continue
                    L.fused_pool, L.out_name, nxt.skip = True, nxt.tops[0], True
        # The box-feature Pooling and MaskPooling (+ its folded Pooling) read the same per-RoI tensor (test.prototxt:571-582 and
        # :631-650): one pass over it (mnc_box_mask_pool) when the mask is ready by the time the Pooling layer runs
        if os.environ.get("MNC_FUSE_POOLS", "1") != "0":
            first_producer = {}
            for i, L in enumerate(self._layers):
                for t in ([L.out_name] if L.out_name else L.tops):
                    first_producer.setdefault(t, i)
            for im, Lm in enumerate(self._layers):
                if Lm.type != "MaskPooling" or not Lm.fused_pool or Lm.skip:
                    continue
                feat, mask = Lm.bottoms[0], Lm.bottoms[1]
                for ib, Lb in enumerate(self._layers[:im]):
                    if (Lb.type == "Pooling" and not Lb.skip and Lb.with_mask is None and Lb.bottoms[0] == feat
                            and self._is_pool2(Lb) and first_producer.get(mask, len(self._layers)) < ib):
                        Lb.with_mask, Lm.skip = Lm, True
                        break
        # residual add folded into the epilogue of the general convolution that produces its second operand:
        # Eltwise SUM (x, conv(...)) [+ ReLU] -> conv writes relu(conv + bias + x) straight into the Eltwise's top
        index = {id(L): i for i, L in enumerate(self._layers)}
        producer = {}
        for L in self._layers:
            if not L.skip:
                for t in L.tops:
                    producer.setdefault(t, L)
        for L in self._layers:
            if L.type != "Eltwise" or len(L.bottoms) != 2 or self._eltwise_op(L) != "SUM":
                continue
            x, y = L.bottoms
            conv = producer.get(y)
            if (conv is not None and conv.type == "Convolution" and self._conv_is_general(conv) and not conv.relu
                    and conv.residual is None and self._consumers.get(y, []) == [index[id(L)]] and y not in self.outputs
                    and x in producer and index[id(producer[x])] < index[id(conv)]):
                conv.residual, conv.out_name, conv.relu, L.skip = x, L.tops[0], L.relu, True
        # sibling InnerProducts on the same bottom without activation (cls_score / seg_cls_score / bbox_pred,
        # test.prototxt:713-785) become ONE GEMM over the concatenated weights; their tops are column slices of it
        by_bottom = {}
        for L in self._layers:
            if L.type == "InnerProduct" and not L.relu and L.act == 0 and not L.skip:
                by_bottom.setdefault(L.bottoms[0], []).append(L)
        for members in by_bottom.values():
            if len(members) > 1:
                members[0].group = members
                for m in members[1:]:
                    m.group_leader = members[0]
        # Two InnerProducts + ReLU of one shape whose inputs both exist when the first one runs -- the box and the mask branch of a
        # head stage (fc6 / fc6_mask, then fc7 / fc7_mask: test.prototxt:584-627 and :652-696; the mask branch's input exists that
        # early because of the one-pass pooling above) -- are ONE launch (mnc_fc_pair: half the K ranges, half the partial sums).
        # The whole-image pipeline pairs the same layers (csrc/pipeline.hip: run_stage), so the two executors keep the same bits.
        # Round 6: fp16 / plain bf16 InnerProducts pair the same way (mnc_fc_lowp_pair).
        if os.environ.get("MNC_FUSE_SMALL", "1") != "0" and self.fc_math in ("fp32", "f16", "bf16", "bf16x3"):
            produced_at = {}
            for i, L in enumerate(self._layers):
                if L.skip:
                    continue
                for t in ([L.out_name] if L.out_name else L.tops):
                    produced_at.setdefault(t, i)
                if L.with_mask is not None:
                    Lm = L.with_mask
                    produced_at.setdefault(Lm.out_name or Lm.tops[0], i)

            def ip_shape(L):                       # (K is not known before the first forward: checked when the pair runs)
                return self._layer_nout(L)
            ips = [(i, L) for i, L in enumerate(self._layers)
                   if L.type == "InnerProduct" and L.relu and not L.skip and L.group is None and L.group_leader is None]
            for a, (i1, L1) in enumerate(ips):
                if L1.pair is not None or L1.pair_leader is not None:
                    continue
                for i2, L2 in ips[a + 1:]:
                    if L2.pair is not None or L2.pair_leader is not None or ip_shape(L2) != ip_shape(L1):
                        continue
                    if produced_at.get(L2.bottoms[0], len(self._layers)) < i1 and L2.bottoms[0] not in L1.tops:
                        L1.pair, L2.pair_leader = L2, L1
                        produced_at[L2.out_name or L2.tops[0]] = i1       # (fc7_mask's input now exists when fc7 runs)
                        break

    def _conv_fast1x1(self, L):
        """A 'general' Convolution that is a plain GEMM: 1x1, no padding, stride 1 or 2 (csrc/conv1x1.hip; MNC_CONV1X1=0 sends
        it through mnc_conv2d like any other geometry).  The fp16 variant walks K in steps of 16 channels."""
        if os.environ.get("MNC_CONV1X1", "1") == "0" or self._conv_kind(L) != "general":
            return False
        k, pad, stride, _, _ = self._conv_geometry(L)
        cin = int(np.asarray(self._layer_weights(L)[0]).shape[1])
        return k == 1 and pad == 0 and stride in (1, 2) and cin % (16 if self.conv_math == "f16" else 8) == 0

    def _plan_formats(self):
        """"f16" math mode: trunk activations travel between the MFMA layers as packed fp16 c8 tensors ('c8h', half the HBM
        bytes; the producer's epilogue rounds exactly as the consumer's staging would).  A Convolution / Pooling writes 'c8h'
        when every reader of its top can take it: the tuned 3x3 kernels, the 1x1 GEMM, MAX pooling, and the residual input of
        a 1x1 GEMM.  Anything else (ROIWarping, the RPN's NCHW heads, Python layers, `.data`) gets fp32 -- written as fp32 by
        the producer, or widened in place by Blob._convert.  MNC_F16_ACTS=0 keeps every tensor fp32."""
        # (round 6: the "bf16x3" / "mixed" and "bf16" modes too -- 'c8x' / 'c8b' between the tuned 3x3 kernels and the MAX 2x2/2
        # poolings, what csrc/pipeline.hip's packed trunk does; the 1x1 GEMM, the general pooling and the stem exist in fp16 only)
        if self.conv_math not in _PACKED_OF or os.environ.get("MNC_F16_ACTS", "1") == "0":
            return
        f16 = self.conv_math == "f16"
        fused_res = {}                      # index of a folded Eltwise -> the convolution that took it over
        for L in self._layers:
            if L.type == "Convolution" and L.residual is not None:
                for i, E in enumerate(self._layers):
                    if E.type == "Eltwise" and E.skip and E.tops[0] == L.out_name:
                        fused_res[i] = L

        def reads_h(i, blob):
            C = self._layers[i]
            if C.type == "Convolution" and not C.skip and C.bottoms[0] == blob:
                return self._conv_kind(C) == "fast3x3" or (f16 and self._conv_fast1x1(C))
            if C.type == "Pooling" and not C.skip:
                if C.msg.get1("pooling_param").get1("pool", "MAX") != "MAX":
                    return False
                return f16 or self._is_pool2(C)
            if i in fused_res:              # read as the residual (the convolution's own output never exists as a blob)
                return f16 and blob == fused_res[i].residual and self._conv_fast1x1(fused_res[i])
            return False

        for L in self._layers:
            if L.skip or L.type != "Convolution":
                continue
            kind = self._conv_kind(L)
            if not (kind in (("c3", "stem", "fast3x3") if f16 else ("c3", "fast3x3")) or (f16 and self._conv_fast1x1(L))):
                continue
            top = L.out_name or L.tops[0]
            cons = self._consumers.get(top, [])
            L.out_h = bool(cons) and top not in self.outputs and all(reads_h(i, top) for i in cons)

    @staticmethod
    def _eltwise_op(L):
        ep = L.msg.get1("eltwise_param")
        op = ep.get1("operation", "SUM") if ep is not None else "SUM"
        if ep is not None and ep.all("coeff"):
            return "COEFF"
        return {1: "SUM", 0: "PROD", 2: "MAX"}.get(op, op)

    @staticmethod
    def _conv_geometry(L):
        """(kernel, pad, stride, num_output, bias_term) of a square, dense, undilated Convolution -- the only kind the kernels
        implement.  Anything else Caffe's convolution_param can express is refused here instead of being run with the wrong
        geometry (dilated conv5 trunks, grouped convolutions, kernel_h != kernel_w ...)."""
        cp = L.msg.get1("convolution_param")
        for key in ("kernel_h", "kernel_w", "pad_h", "pad_w", "stride_h", "stride_w"):
            if cp.get1(key) is not None:
                raise NotImplementedError("Convolution %s: %s is not supported (square kernel_size / pad / stride only)"
                                          % (L.name, key))
        if cp.get1("dilation", 1) != 1:
            raise NotImplementedError("Convolution %s: dilation %r is not supported" % (L.name, cp.get1("dilation")))
        if cp.get1("group", 1) != 1:
            raise NotImplementedError("Convolution %s: group %r is not supported" % (L.name, cp.get1("group")))
        if cp.get1("kernel_size") is None:
            raise NotImplementedError("Convolution %s: kernel_size is required" % L.name)
        return cp.get1("kernel_size"), cp.get1("pad", 0), cp.get1("stride", 1), cp.get1("num_output"), cp.get1("bias_term", True)

    def _conv_kind(self, L):
        """Which kernel family runs this Convolution: 'c3' / 'stem' (3-channel NCHW input blob), 'fast3x3' (the tuned 3x3 pad 1
        stride 1 kernels), 'nchw1x1' (the RPN's 1x1 heads, which feed Reshape / Python layers in NCHW) or 'general' (mnc_conv2d)."""
        k, pad, stride, cout, _ = self._conv_geometry(L)
        first = L.bottoms[0] in self.inputs
        if first:
            return "c3" if (k == 3 and pad == 1 and stride == 1) else "stem"
        if k == 3 and pad == 1 and stride == 1 and cout % 32 == 0 and L.residual is None:
            return "fast3x3"
        if k == 1 and pad == 0 and stride == 1 and not L.relu and L.bn is None and L.scale is None and L.residual is None \
                and self._rpn_head_conv(L):
            return "nchw1x1"
        return "general"

    def _conv_is_general(self, L):
        return self._conv_kind(L) == "general"

    def _rpn_head_conv(self, L):
        """1x1 convolutions whose consumers want NCHW (Reshape / Python layers): rpn_cls_score, rpn_bbox_pred."""
        cons = [self._layers[i].type for i in self._consumers.get(L.tops[0], [])]
        return bool(cons) and all(t in ("Reshape", "Python", "Softmax") for t in cons)

    @staticmethod
    def _is_pool2(L):
        p = L.msg.get1("pooling_param")
        return (p.get1("pool", "MAX") == "MAX" and p.get1("kernel_size") == 2 and p.get1("stride") == 2
                and p.get1("pad", 0) == 0)

    # ------------------------------------------------------------------------------------------------ binding
    def _bind_layers(self):
        for i, L in enumerate(self._layers):
            if L.skip:
                continue
            binder = getattr(self, "_bind_" + L.type, None)
            if binder is None:
                raise NotImplementedError("layer type %r (%s) is not supported by the MI355X engine" % (L.type, L.name))
            L.run = binder(L, i)
        # static channel/geometry facts the binders need are discovered lazily at first forward

    def _h(self):
        return self._ctx.h

    def _folded_conv_params(self, L):
        """(W, b) of a Convolution with its in-place BatchNorm (use_global_stats: mean / var / moving-average factor blobs) and
        Scale (gamma, beta) folded in: y = gamma * (conv(x) + b - mean) / sqrt(var + eps) + beta.  Folded in float64."""
        _, _, _, cout, _ = self._conv_geometry(L)
        blobs = self._layer_weights(L)
        W = np.asarray(blobs[0], dtype=F32)
        b = np.asarray(blobs[1], dtype=F32) if len(blobs) > 1 and blobs[1] is not None else np.zeros(cout, F32)
        self.params[L.name] = [_Param(W), _Param(b)] if len(blobs) > 1 and blobs[1] is not None else [_Param(W)]
        if L.bn is None and L.scale is None:
            return W, b
        a, c = np.ones(cout, np.float64), np.zeros(cout, np.float64)       # y = a * conv_out + c, per channel
        if L.bn is not None:
            mean, var, factor = [np.asarray(x, dtype=np.float64) for x in self._layer_weights(L.bn)[:3]]
            bp = L.bn.msg.get1("batch_norm_param")
            eps = float(bp.get1("eps", 1e-5)) if bp is not None else 1e-5
            f = float(factor.reshape(-1)[0])
            inv = 0.0 if f == 0.0 else 1.0 / f
            a = 1.0 / np.sqrt(var.reshape(-1) * inv + eps)
            c = -mean.reshape(-1) * inv * a
            self.params[L.bn.name] = [_Param(np.asarray(x, dtype=F32)) for x in self._layer_weights(L.bn)[:3]]
        if L.scale is not None:
            sw = self._layer_weights(L.scale)
            gamma = np.asarray(sw[0], dtype=np.float64).reshape(-1)
            beta = np.asarray(sw[1], dtype=np.float64).reshape(-1) if len(sw) > 1 and sw[1] is not None else np.zeros(cout)
            a, c = a * gamma, c * gamma + beta
            self.params[L.scale.name] = [_Param(np.asarray(x, dtype=F32)) for x in sw if x is not None]
        Wf = (W.astype(np.float64) * a[:, None, None, None]).astype(F32)
        bf = (b.astype(np.float64) * a + c).astype(F32)
        return Wf, bf

    def _bind_Convolution(self, L, i):
        k, pad, stride, cout, _ = self._conv_geometry(L)
        kind = self._conv_kind(L)
        W, b = self._folded_conv_params(L)
        cin = W.shape[1]
        key = tuple(L.param_names) if all(L.param_names) and L.param_names else (L.name,)
        d_b = self._dev_param(key + ("b",), lambda: self._upload(b))
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.out_name or L.tops[0]]
        relu = 1 if L.relu else 0
        if kind == "c3":
            d_w = self._dev_param(key + ("w",), lambda: self._upload(W))

            def run():
                N, _, H, Wd = bot.shape
                src = bot.dev_in("plain")
                top.reshape(N, cout, H, Wd)
                pk = _PACKED_OF.get(self.conv_math, "c8h")
                dst = top.dev_out(pk if L.out_h else "c8")
                ob = _PACKED[pk][0] if L.out_h else 4
                for n in range(N):                       # one launch sequence per image of the batch
                    if L.out_h:                          # (out_fmt of mnc_conv3x3_c3_fmt: 1 split bf16, 2 fp16, 3 bf16)
                        _lib.call("mnc_conv3x3_c3_fmt", self._h(), src + n * 3 * H * Wd * 4, d_w, d_b,
                                  dst + n * cout * H * Wd * ob, H, Wd, cout, relu, _PACKED[pk][1] + 1)
                    else:
                        _lib.call("mnc_conv3x3_c3", self._h(), src + n * 3 * H * Wd * 4, d_w, d_b, dst + n * cout * H * Wd * 4,
                                  H, Wd, cout, relu)
            return run
        if kind == "stem":
            if cin != 3 or W.shape[2] != W.shape[3]:
                raise NotImplementedError("Convolution %s on the input blob: 3 channels, square kernel" % L.name)
            # "f16" mode: the stem on the fp16 matrix pipe like every other convolution of the mode (kernel rows padded to 8 taps,
            # weights in registers; csrc/conv_gen.hip) -- MNC_STEM_F16=0 keeps the fp32 VALU kernel
            mfma = (self.conv_math == "f16" and k in (3, 5, 7) and cout % 32 == 0 and os.environ.get("MNC_STEM_F16", "1") != "0")
            if mfma:
                def build_stem():
                    raw = self._upload(W)
                    packed = self._ctx.alloc(((3 * k + 1) // 2) * (cout // 32) * 1024)
                    _lib.call("mnc_pack_conv_stem_f16", self._h(), raw, packed, cout, k)
                    self._ctx.free(raw)
                    return packed
                d_w = self._dev_param(key + ("w", "stem_f16"), build_stem)
            else:
                d_w = self._dev_param(key + ("w",), lambda: self._upload(W))

            def run():
                N, _, H, Wd = bot.shape
                OH, OW = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
                src = bot.dev_in("plain")
                top.reshape(N, cout, OH, OW)
                dst = top.dev_out("c8h" if L.out_h else "c8")
                ob = 2 if L.out_h else 4
                for n in range(N):
                    _lib.call("mnc_conv_stem_f16" if mfma else "mnc_conv_stem_c3_fmt", self._h(), src + n * 3 * H * Wd * 4, d_w, d_b,
                              dst + n * cout * OH * OW * ob, H, Wd, cout, k, stride, pad, relu, 1 if L.out_h else 0)
            return run
        if kind == "fast3x3":
            x3 = self.conv_math in ("bf16x3", "f16", "bf16")
            # fp32 mode: the direct implicit GEMM, or Winograd F(2x2,3x3) on the same fp32 matrix pipe (2.25x fewer multiplies,
            # equal to the direct form up to fp32 rounding; MNC_CONV_WINOGRAD / Net(winograd=))
            pitch, pack, conv = (84, "mnc_pack_conv3x3_f16", "mnc_conv3x3_f16") if self.conv_math == "f16" else \
                                (84, "mnc_pack_conv3x3_bf16", "mnc_conv3x3_bf16") if self.conv_math == "bf16" else \
                                (84, "mnc_pack_conv3x3_bf16x3", "mnc_conv3x3_bf16x3") if x3 else \
                                (288, "mnc_pack_conv3x3_wino4", "mnc_conv3x3_wino4") if self._wino == 4 else \
                                (136, "mnc_pack_conv3x3_wino", "mnc_conv3x3_wino") if self._wino == 2 else \
                                (76, "mnc_pack_conv3x3_weights", "mnc_conv3x3")

            def build():
                raw = self._upload(W)
                nbytes = (cin // 8) * cout * pitch * 4
                if x3:     # include/mnc_hip.h: mnc_conv3x3_lowp_weight_bytes (bf16x3: four planes per 16-channel block, else two)
                    nbytes = max(nbytes, ((cin + 15) // 16) * (cout // 32) * (4 if self.conv_math == "bf16x3" else 2) * 4608)
                packed = self._ctx.alloc(nbytes)
                _lib.call(pack, self._h(), raw, packed, cout, cin)
                self._ctx.free(raw)
                return packed
            d_w = self._dev_param(key + ("w", self.conv_math, ("wino%d" % self._wino) if (self._winograd and not x3) else "direct"), build)

            def run():
                N, _, H, Wd = bot.shape
                pk = _PACKED_OF.get(self.conv_math)
                in_h = pk is not None and bot._dev_valid and bot.layout == pk
                src = bot.dev_in(pk if in_h else "c8")
                if L.fused_pool and pk is not None:        # reduced precision: packed in, pooled packed out (a fp32 consumer widens it)
                    if not in_h:
                        src = bot.dev_in(pk)
                    OH, OW = _pool_out(H), _pool_out(Wd)
                    top.reshape(N, cout, OH, OW)
                    dst = top.dev_out(pk)
                    eb, mode = _PACKED[pk]
                    for n in range(N):
                        _lib.call("mnc_conv3x3_lowp_pool", self._h(), mode, src + n * cin * H * Wd * eb, d_w, d_b,
                                  dst + n * cout * OH * OW * eb, H, Wd, cin, cout, relu)
                    return
                if L.fused_pool:                           # top is the Pooling layer's blob
                    OH, OW = _pool_out(H), _pool_out(Wd)
                    top.reshape(N, cout, OH, OW)
                    dst = top.dev_out("c8")
                    for n in range(N):
                        _lib.call("mnc_conv3x3_wino4_pool" if self._wino == 4 else "mnc_conv3x3_wino_pool", self._h(), src + n * cin * H * Wd * 4, d_w, d_b,
                                  dst + n * cout * OH * OW * 4, H, Wd, cin, cout, relu)
                    return
                top.reshape(N, cout, H, Wd)
                if in_h or L.out_h:                        # packed activation tensors on either side (the mode's own form)
                    dst = top.dev_out(pk if L.out_h else "c8")
                    eb = _PACKED[pk][0]
                    ib, ob = (eb if in_h else 4), (eb if L.out_h else 4)
                    for n in range(N):
                        _lib.call(conv + "_pk", self._h(), src + n * cin * H * Wd * ib, d_w, d_b,
                                  dst + n * cout * H * Wd * ob, H, Wd, cin, cout, relu, 1 if in_h else 0, 1 if L.out_h else 0)
                    return
                dst = top.dev_out("c8")
                for n in range(N):
                    _lib.call(conv, self._h(), src + n * cin * H * Wd * 4, d_w, d_b, dst + n * cout * H * Wd * 4, H, Wd, cin,
                              cout, relu)
            return run
        if kind == "nchw1x1":
            d_w = self._dev_param(key + ("w",), lambda: self._upload(W.reshape(cout, cin)))

            def run():
                N, _, H, Wd = bot.shape
                src = bot.dev_in("c8")
                top.reshape(N, cout, H, Wd)
                dst = top.dev_out("plain")
                for n in range(N):
                    _lib.call("mnc_conv1x1_to_nchw", self._h(), src + n * cin * H * Wd * 4, d_w, d_b,
                              dst + n * cout * H * Wd * 4, H, Wd, cin, cout)
            return run
        # general: any kernel / stride / pad on the fp32 matrix pipe (fp16 pipe in the f16 mode), residual add and ReLU in the epilogue
        if cin % 8 or cout % 8:
            raise NotImplementedError("Convolution %s: channel counts must be multiples of 8 (got %d -> %d)" % (L.name, cin, cout))
        kh, kw = W.shape[2], W.shape[3]

        f16 = self.conv_math == "f16"
        conv2d = "mnc_conv2d_f16" if f16 else "mnc_conv2d"
        if self._conv_fast1x1(L):
            return self._bind_conv1x1(L, W, d_b, key, bot, top, stride, relu)

        def build_general():
            raw = self._upload(W)
            packed = self._ctx.alloc(kh * kw * ((cin + 31) // 32) * 32 * cout * 2 if f16 else W.nbytes)
            _lib.call("mnc_pack_conv_weights_f16" if f16 else "mnc_pack_conv_weights", self._h(), raw, packed, cout, cin, kh, kw)
            self._ctx.free(raw)
            return packed
        d_w = self._dev_param(key + ("w", "general", "f16" if f16 else "fp32"), build_general)
        res = self.blobs[L.residual] if L.residual else None

        def run():
            N, _, H, Wd = bot.shape
            OH, OW = (H + 2 * pad - kh) // stride + 1, (Wd + 2 * pad - kw) // stride + 1
            src = bot.dev_in("c8")
            rsrc = res.dev_in("c8") if res is not None else None
            if res is not None and tuple(res.shape) != (N, cout, OH, OW):
                raise ValueError("Convolution %s: residual %r has shape %r, expected %r" % (L.name, res.name, res.shape,
                                                                                            (N, cout, OH, OW)))
            top.reshape(N, cout, OH, OW)
            dst = top.dev_out("c8")
            for n in range(N):
                _lib.call(conv2d, self._h(), src + n * cin * H * Wd * 4, d_w, d_b,
                          (rsrc + n * cout * OH * OW * 4) if rsrc is not None else None, dst + n * cout * OH * OW * 4, H, Wd, cin,
                          cout, kh, kw, stride, pad, relu)
        return run

    def _bind_conv1x1(self, L, W, d_b, key, bot, top, stride, relu):
        """1x1 convolution (stride 1 / 2) as a plain GEMM straight from the c8 tensors (csrc/conv1x1.hip).  fp32 / bf16x3 modes:
        fp32 matrix pipe, fp32 tensors.  f16 mode: packed fp16 input (converted once if a producer left fp32), packed or fp32
        output as planned (_plan_formats), residual in whichever form its producer wrote."""
        cout, cin = int(W.shape[0]), int(W.shape[1])
        f16 = self.conv_math == "f16"

        def build():
            raw = self._upload(W.reshape(cout, cin))
            packed = self._ctx.alloc(cin * ((cout + 31) // 32) * 32 * (2 if f16 else 4))
            _lib.call("mnc_pack_conv1x1", self._h(), raw, packed, cout, cin, 1 if f16 else 0)
            self._ctx.free(raw)
            return packed
        d_w = self._dev_param(key + ("w", "1x1", "f16" if f16 else "fp32"), build)
        res = self.blobs[L.residual] if L.residual else None

        def run():
            N, _, H, Wd = bot.shape
            OH, OW = (H - 1) // stride + 1, (Wd - 1) // stride + 1
            if res is not None and tuple(res.shape) != (N, cout, OH, OW):
                raise ValueError("Convolution %s: residual %r has shape %r, expected %r" % (L.name, res.name, res.shape,
                                                                                            (N, cout, OH, OW)))
            if not f16:
                src = bot.dev_in("c8")
                rsrc = res.dev_in("c8") if res is not None else None
                top.reshape(N, cout, OH, OW)
                dst = top.dev_out("c8")
                for n in range(N):
                    _lib.call("mnc_conv1x1", self._h(), src + n * cin * H * Wd * 4, d_w, d_b,
                              (rsrc + n * cout * OH * OW * 4) if rsrc is not None else None, dst + n * cout * OH * OW * 4,
                              H, Wd, cin, cout, stride, relu)
                return
            src = bot.dev_in("c8h")
            res_h = res is not None and res._dev_valid and res.layout == "c8h"
            rsrc = res.dev_in("c8h" if res_h else "c8") if res is not None else None
            rb, ob = (2 if res_h else 4), (2 if L.out_h else 4)
            top.reshape(N, cout, OH, OW)
            dst = top.dev_out("c8h" if L.out_h else "c8")
            for n in range(N):
                _lib.call("mnc_conv1x1_f16_pk", self._h(), src + n * cin * H * Wd * 2, d_w, d_b,
                          (rsrc + n * cout * OH * OW * rb) if rsrc is not None else None, dst + n * cout * OH * OW * ob,
                          H, Wd, cin, cout, stride, relu, 1 if res_h else 0, 1 if L.out_h else 0)
        return run

    def _bind_ReLU(self, L, i):
        raise NotImplementedError("stand-alone ReLU %s (only in-place ReLU after Convolution/InnerProduct)" % L.name)

    def _bind_Dropout(self, L, i):
        """Identity at test time (Caffe scales at train time; faster_rcnn_end2end/test.prototxt:546-555)."""
        if L.tops[0] == L.bottoms[0]:
            return None
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]

        def run():
            src = bot.dev_in("plain")
            top.reshape(*bot.shape)
            if bot.count:
                _lib.call("mnc_copy2d", self._h(), top.dev_out("plain"), bot.shape[-1], src, bot._ld() if len(bot.shape) == 2
                          else bot.shape[-1], bot.count // bot.shape[-1], bot.shape[-1])
            else:
                top.dev_out("plain")
        return run

    def _bind_Pooling(self, L, i):
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]
        if not self._is_pool2(L):
            p = L.msg.get1("pooling_param")
            if p.get1("pool", "MAX") != "MAX":
                raise NotImplementedError("Pooling %s: only MAX" % L.name)
            k, stride, pad = p.get1("kernel_size"), p.get1("stride", 1), p.get1("pad", 0)

            def out_size(n):                       # Caffe: ceil, and the last window must start inside the image
                o = -(-(n + 2 * pad - k) // stride) + 1
                return o - 1 if pad > 0 and (o - 1) * stride >= n + pad else o

            def run_general():
                N, C, H, W = bot.shape
                half = bot._dev_valid and bot.layout == "c8h"       # packed fp16 in -> packed fp16 out (max commutes with rounding)
                lay, eb = ("c8h", 2) if half else ("c8", 4)
                src = bot.dev_in(lay)
                OH, OW = out_size(H), out_size(W)
                top.reshape(N, C, OH, OW)
                dst = top.dev_out(lay)
                for n in range(N):
                    _lib.call("mnc_maxpool_c8_f16" if half else "mnc_maxpool_c8", self._h(), src + n * C * H * W * eb,
                              dst + n * C * OH * OW * eb, C, H, W, k, stride, pad)
            return run_general

        def run():
            if bot._dev_valid and (bot.layout == "c8" or bot.layout in _PACKED):
                N, C, H, W = bot.shape
                lay = bot.layout                           # packed in -> packed out (max commutes with the rounding / the split)
                eb = _PACKED[lay][0] if lay in _PACKED else 4
                fn = {"c8": "mnc_maxpool2_c8", "c8h": "mnc_maxpool2_c8_f16", "c8x": "mnc_maxpool2_c8_bf16x3", "c8b": "mnc_maxpool2_c8_bf16"}[lay]
                src = bot.dev_in(lay)
                OH, OW = _pool_out(H), _pool_out(W)
                top.reshape(N, C, OH, OW)
                dst = top.dev_out(lay)
                for n in range(N):
                    _lib.call(fn, self._h(), src + n * C * H * W * eb, dst + n * C * OH * OW * eb, C, H, W)
            else:
                R, C, PH, PW = bot.shape
                if PH % 2 or PW % 2:            # Caffe's ceil rule would give (PH + 1) // 2 (a 7x7 input -> 4x4): no kernel for it
                    raise NotImplementedError("Pooling %s on per-RoI features: odd size %dx%d (MAX 2x2/2 needs even sizes)"
                                              % (L.name, PH, PW))
                top.reshape(R, C, PH // 2, PW // 2)
                K = C * (PH // 2) * (PW // 2)
                fmt = self._sm_format(top.name, R, K, C) if R else 0
                if L.with_mask is not None:             # + MaskPooling and its Pooling of the same tensor, same pass
                    mtop = self.blobs[L.with_mask.out_name]
                    d_mask = self.blobs[L.with_mask.bottoms[1]].dev_in("plain")
                    mtop.reshape(R, C, PH // 2, PW // 2)
                    mfmt = self._sm_format(mtop.name, R, K, C) if R else 0
                    if R and fmt == mfmt and C % 8 == 0:
                        # round 6: the tensor is read from its stage-major copy when its producer wrote one (what
                        # csrc/pipeline.hip's run_stage does: mnc_hip.h, mnc_box_mask_pool_ex -- same bits in both executors), and
                        # the pooled tensors are written in the stage-major form only when nothing else reads them
                        sm_in = bot._sm if (fmt and bot._sm and bot._sm["fmt"] in (1, 2, 3) and bot._sm["M"] == R
                                            and bot._sm["K"] == C * PH * PW and (bot._dev_valid or bot._sm_only)) else None
                        src = None if (sm_in and not bot._dev_valid) else bot.dev_in("rhwc")
                        sm_only = bool(fmt) and self._sm_only_ok(top.name) and self._sm_only_ok(mtop.name)
                        dst = top.dev_out_sm_only("rhwc") if sm_only else top.dev_out("rhwc")
                        mdst = mtop.dev_out_sm_only("rhwc") if sm_only else mtop.dev_out("rhwc")
                        _lib.call("mnc_box_mask_pool_ex", self._h(), src, sm_in["ptr"] if sm_in else None, sm_in["fmt"] if sm_in else 0, d_mask,
                                  dst, mdst, R, PH, PW, C, top.sm_out(fmt, R, K) if fmt else None,
                                  mtop.sm_out(fmt, R, K) if fmt else None, fmt)
                        return
                    src = bot.dev_in("rhwc")
                    dst = top.dev_out("rhwc")
                    mdst = mtop.dev_out("rhwc")
                    if R:
                        if mfmt:
                            _lib.call("mnc_mask_pool_sm", self._h(), src, d_mask, mdst, R, PH, PW, C, 1, mtop.sm_out(mfmt, R, K), mfmt)
                        else:
                            _lib.call("mnc_mask_pool", self._h(), src, d_mask, mdst, R, PH, PW, C, 1)
                else:
                    src = bot.dev_in("rhwc")
                    dst = top.dev_out("rhwc")
                if fmt:
                    _lib.call("mnc_maxpool2_rhwc_sm", self._h(), src, dst, R, PH, PW, C, top.sm_out(fmt, R, K), fmt)
                elif R:
                    _lib.call("mnc_maxpool2_rhwc", self._h(), src, dst, R, PH, PW, C)
        return run

    def _bind_Eltwise(self, L, i):
        """Eltwise SUM of two feature maps (+ folded in-place ReLU): ResNet's residual add when it was not folded into the
        producing convolution."""
        if self._eltwise_op(L) != "SUM" or len(L.bottoms) != 2:
            raise NotImplementedError("Eltwise %s: only SUM of two bottoms without coefficients" % L.name)
        a, b, top = self.blobs[L.bottoms[0]], self.blobs[L.bottoms[1]], self.blobs[L.tops[0]]
        relu = 1 if L.relu else 0

        def run():
            if tuple(a.shape) != tuple(b.shape):
                raise ValueError("Eltwise %s: shapes %r and %r differ" % (L.name, a.shape, b.shape))
            layout = "c8" if len(a.shape) == 4 and a.shape[1] % 8 == 0 and (a._dev_valid and (a.layout == "c8" or a.layout in _PACKED)) else "plain"
            pa, pb = a.dev_in(layout), b.dev_in(layout)
            top.reshape(*a.shape)
            _lib.call("mnc_add", self._h(), pa, pb, top.dev_out(layout), a.count, relu)
        return run

    def _bind_Reshape(self, L, i):
        # only as part of Reshape(0,2,-1,0) -> Softmax(axis 1) -> Reshape(0,2A,-1,0)  (test.prototxt:440-462)
        dims = L.msg.get1("reshape_param").get1("shape").all("dim")
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]

        def run():
            shp = list(bot.shape)
            out = [shp[k] if d == 0 else d for k, d in enumerate(dims)]
            if -1 in out:
                known = int(np.prod([d for d in out if d != -1]))
                out[out.index(-1)] = bot.count // known
            src = bot.dev_in("plain")
            top.reshape(*out)
            _lib.call("mnc_d2d", self._h(), top.dev_out("plain"), src, bot.count * 4)
        return run

    def _bind_Softmax(self, L, i):
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]

        def run():
            src = bot.dev_in("plain")
            top.reshape(*bot.shape)
            if len(bot.shape) == 2:
                if bot.shape[0]:
                    _lib.call("mnc_softmax_rows_ld", self._h(), src, bot._ld(), top.dev_out("plain"), bot.shape[0],
                              bot.shape[1])
                else:
                    top.dev_out("plain")
            elif len(bot.shape) == 4 and bot.shape[0] == 1 and bot.shape[1] == 2:
                # (1, 2, A*H, W): channel a pairs with channel A + a of the un-reshaped score blob
                _lib.call("mnc_rpn_softmax", self._h(), src, top.dev_out("plain"), 1, bot.shape[2], bot.shape[3])
            else:
                raise NotImplementedError("Softmax %s over shape %r" % (L.name, bot.shape))
        return run

    def _bind_Sigmoid(self, L, i):
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]

        def run():
            src = bot.dev_in("plain")
            top.reshape(*bot.shape)
            _lib.call("mnc_eltwise", self._h(), src, top.dev_out("plain"), bot.count, 2)
        return run

    def _sm_format(self, blob, M, K, C):
        """Second output of a per-RoI producer (ROIWarping / Pooling / MaskPooling) -> 0 none, 1 f16, 2 bf16x3, 3 bf16: the stage-major
        2-byte form (Blob._sm) when an InnerProduct that reads `blob` will run on the reduced-precision kernels for this M
        (same rule as _bind_InnerProduct.weights_for), so that it does not have to convert the fp32 rows itself.  MNC_FC_SM=0
        switches the second outputs off."""
        if self.fc_math == "fp32" or M <= 0 or os.environ.get("MNC_FC_SM", "1") == "0":
            return 0
        fmt = 0
        for i in self._consumers.get(blob, []):
            L = self._layers[i]
            if L.type != "InnerProduct" or L.skip or L.group is not None or L.group_leader is not None:
                continue
            n_out = L.msg.get1("inner_product_param").get1("num_output")
            if 2.0 * M * n_out * K < _X3_MIN_FLOPS:
                continue
            fm = "bf16x3" if (self.math == "mixed" and K > 50000) else self.fc_math      # (weights_for's rule)
            if fm == "f16" and K % 64 == 0 and C % 64 == 0:
                fmt = 1
            elif fm == "bf16":                            # round 6: format 3 = fp16's layout, bf16 values (mnc_fc_bf16_ex)
                if K % 64 == 0 and C % 64 == 0:
                    fmt = 3
            elif K % 32 == 0 and C % 32 == 0 and not (fm == "f16" and K % 64 == 0):
                fmt = 2
        return fmt

    def _sm_only_ok(self, blob):
        """May a per-RoI tensor exist in its stage-major form only?  Every reader must take that form: the reduced-precision
        InnerProducts (they multiply from it) and the one-pass box / mask pooling (mnc_box_mask_pool_ex reads it).  Anything
        else -- or `.data` -- still works (Blob._materialize), it just pays a conversion pass, so the answer is 'no' then."""
        if blob in self.outputs or os.environ.get("MNC_SM_ONLY", "1") == "0":
            return False
        cons = self._consumers.get(blob, [])
        if not cons:
            return False
        for i in cons:
            L = self._layers[i]
            if L.type == "InnerProduct" and not L.skip and L.group is None and L.group_leader is None:
                continue
            if L.type == "Pooling" and not L.skip and L.with_mask is not None:
                continue
            if L.type == "MaskPooling" and L.skip:          # folded into the Pooling above
                continue
            return False
        return True

    def _bind_ROIWarping(self, L, i):
        rp = L.msg.get1("roi_warping_param")
        ph, pw, scale = rp.get1("pooled_h"), rp.get1("pooled_w"), float(rp.get1("spatial_scale"))
        feat, rois = self.blobs[L.bottoms[0]], self.blobs[L.bottoms[1]]
        top = self.blobs[L.out_name or L.tops[0]]
        pool2 = 1 if L.fused_pool else 0
        oh, ow = (ph // 2, pw // 2) if pool2 else (ph, pw)

        def run():
            N, C, H, W = feat.shape
            if N != 1:
                raise NotImplementedError("ROIWarping %s: one image per forward (got a batch of %d)" % (L.name, N))
            R = rois.shape[0]
            d_feat, d_rois = feat.dev_in("c8"), rois.dev_in("plain")
            top.reshape(R, C, oh, ow)
            fmt = self._sm_format(top.name, R, C * oh * ow, C)
            sm_only = False
            if fmt and R and (C * oh * ow) % 64 == 0 and self._sm_only_ok(top.name):
                ok = ctypes.c_int(0)
                _lib.call("mnc_roi_warp_sm_only_ok", self._h(), C, pool2, ctypes.addressof(ok))
                sm_only = bool(ok.value)
            dst = top.dev_out_sm_only("rhwc") if sm_only else top.dev_out("rhwc")
            if fmt:
                _lib.call("mnc_roi_warp_sm", self._h(), d_feat, C, H, W, d_rois, R, oh, ow, scale, pool2, dst,
                          top.sm_out(fmt, R, C * oh * ow), fmt)
            else:
                _lib.call("mnc_roi_warp", self._h(), d_feat, C, H, W, d_rois, R, oh, ow, scale, pool2, dst)
        return run

    def _bind_ROIPooling(self, L, i):
        """Fast R-CNN RoI max pooling over a batch of images (CFM: models/VGG16/cfm/test.prototxt:397-407, 446-456)."""
        rp = L.msg.get1("roi_pooling_param")
        ph, pw, scale = rp.get1("pooled_h"), rp.get1("pooled_w"), float(rp.get1("spatial_scale"))
        feat, rois = self.blobs[L.bottoms[0]], self.blobs[L.bottoms[1]]
        top = self.blobs[L.tops[0]]

        def run():
            N, C, H, W = feat.shape
            R = rois.shape[0]
            d_feat, d_rois = feat.dev_in("c8"), rois.dev_in("plain")
            top.reshape(R, C, ph, pw)
            _lib.call("mnc_roi_pool", self._h(), d_feat, N, C, H, W, d_rois, R, ph, pw, scale, top.dev_out("rhwc"))
        return run

    def _bind_MaskResize(self, L, i):
        mp = L.msg.get1("mask_resize_param")
        oh, ow = mp.get1("output_height"), mp.get1("output_width")
        bot, top = self.blobs[L.bottoms[0]], self.blobs[L.tops[0]]

        def run():
            R, C, IH, IW = bot.shape
            src = bot.dev_in("plain")
            top.reshape(R, C, oh, ow)
            _lib.call("mnc_mask_resize", self._h(), src, top.dev_out("plain"), R * C, IH, IW, oh, ow)
        return run

    def _bind_MaskPooling(self, L, i):
        feat, mask = self.blobs[L.bottoms[0]], self.blobs[L.bottoms[1]]
        top = self.blobs[L.out_name or L.tops[0]]
        pool2 = 1 if L.fused_pool else 0

        def run():
            R, C, PH, PW = feat.shape
            if pool2 and (PH % 2 or PW % 2):
                raise NotImplementedError("MaskPooling %s + MAX 2x2/2 fused: odd size %dx%d" % (L.name, PH, PW))
            d_feat, d_mask = feat.dev_in("rhwc"), mask.dev_in("plain")
            oh, ow = (PH // 2, PW // 2) if pool2 else (PH, PW)
            top.reshape(R, C, oh, ow)
            dst = top.dev_out("rhwc")
            fmt = self._sm_format(top.name, R, C * oh * ow, C)
            if fmt:
                _lib.call("mnc_mask_pool_sm", self._h(), d_feat, d_mask, dst, R, PH, PW, C, pool2, top.sm_out(fmt, R, C * oh * ow),
                          fmt)
            else:
                _lib.call("mnc_mask_pool", self._h(), d_feat, d_mask, dst, R, PH, PW, C, pool2)
        return run

    def _bind_InnerProduct(self, L, i):
        if L.group_leader is not None:                 # computed by the group's leader
            W, b = self._layer_weights(L)
            self.params[L.name] = [_Param(W), _Param(b)]
            return None
        if L.group is not None:
            return self._bind_ip_group(L)
        n_out = L.msg.get1("inner_product_param").get1("num_output")
        W, b = self._layer_weights(L)
        self.params[L.name] = [_Param(W), _Param(b)]
        key = tuple(L.param_names) if L.param_names and all(L.param_names) else (L.name,)
        d_b = self._dev_param(key + ("b",), lambda: self._upload(b))
        bot = self.blobs[L.bottoms[0]]
        top = self.blobs[L.out_name or L.tops[0]]
        act = 1 if L.relu else L.act
        K = W.shape[1]
        state = {}

        def weights_for(shape, M):
            """Caffe flattens (C,PH,PW); the engine's per-RoI features are (PH,PW,C): permute the columns once.  In
            bf16x3 mode the large products additionally get the weights pre-split into hi/lo bf16."""
            big = 2.0 * M * n_out * K >= _X3_MIN_FLOPS
            # mixed: an InnerProduct over more than 50 000 inputs (fc6_maskest: 100 352) stays split-bf16 (pipeline.hip: prepare_fc)
            fm = "bf16x3" if (self.math == "mixed" and K > 50000) else self.fc_math
            bf = fm == "bf16" and K % 64 == 0 and big          # plain bf16: the fp16 kernel's layout and launcher
            f16 = (fm == "f16" and K % 64 == 0 and big) or bf
            x3 = (not f16) and fm in ("bf16x3", "f16") and K % 32 == 0 and big
            tag = ("bf16",) if bf else ("f16",) if f16 else ("x3",) if x3 else ()
            fn = "mnc_fc_bf16" if bf else "mnc_fc_f16" if f16 else "mnc_fc_bf16x3" if x3 else "mnc_fc"

            def finish(d_w):
                if not (x3 or f16):
                    return d_w
                packed = self._ctx.alloc((n_out + 127) // 128 * 128 * K * (2 if f16 else 4))
                _lib.call("mnc_pack_fc_bf16" if bf else "mnc_pack_fc_f16" if f16 else "mnc_pack_fc_bf16x3", self._h(), d_w, packed, n_out, K)
                self._ctx.free(d_w)
                return packed

            if len(shape) == 4 and shape[2] * shape[3] > 1:
                geo = (shape[1], shape[2], shape[3])

                def build():
                    raw = self._upload(W)
                    packed = self._ctx.alloc(W.nbytes)
                    _lib.call("mnc_pack_fc_weights", self._h(), raw, packed, n_out, geo[0], geo[1], geo[2])
                    self._ctx.free(raw)
                    return finish(packed)
                return self._dev_param(key + ("w",) + geo + tag, build), "rhwc", fn
            return self._dev_param(key + ("w", "plain") + tag, lambda: finish(self._upload(W))), "plain", fn

        LOWP_PAIR = {"mnc_fc_bf16x3": 0, "mnc_fc_f16": 1, "mnc_fc_bf16": 2}          # mnc_fc_lowp_pair's mode per single-call entry point

        def lowp_args(M):
            """(fp32 rows or None, stage-major rows or None, dst, second-output format, second output or None) of this layer's
            reduced-precision call, with its top made ready: the rows arrive in the kernel's own 2-byte form when the producer
            wrote them (Blob._sm: no conversion pass), and leave in the NEXT InnerProduct's form as well when one will read them."""
            want = {"mnc_fc_f16": 1, "mnc_fc_bf16x3": 2, "mnc_fc_bf16": 3}.get(state.get("fn"), 0)
            sm = bot._sm
            pre = bool(want and sm is not None and sm["fmt"] == want and sm["M"] == M and sm["K"] == K
                       and (bot._dev_valid or bot._sm_only) and bot.layout == state["layout"])
            src = None if pre else bot.dev_in(state["layout"])
            top.reshape(M, n_out)
            dst = top.dev_out("plain")
            ofmt = self._sm_format(top.name, M, n_out, n_out) if (want and top._view is None) else 0
            return src, (sm["ptr"] if pre else None), dst, ofmt, (top.sm_out(ofmt, M, n_out) if ofmt else None)

        def run():
            M = bot.shape[0]
            if int(np.prod(bot.shape[1:])) != K:
                raise ValueError("InnerProduct %s: input %r does not flatten to K=%d" % (L.name, bot.shape, K))
            if L.pair_done:                        # computed with its pair leader earlier in this forward
                L.pair_done = False
                return
            if M and "w" not in state:
                state["w"], state["layout"], state["fn"] = weights_for(bot.shape, M)
            P = L.pair
            if M and state.get("fn") in LOWP_PAIR and P is not None and P.ip_prepare is not None:
                src, pre, dst, ofmt, osm = lowp_args(M)
                other = P.ip_prepare(M, K, state["fn"])
                if other is not None and other["ofmt"] == ofmt and other["ld"] == top._ld():
                    _lib.call("mnc_fc_lowp_pair", self._h(), LOWP_PAIR[state["fn"]], src, pre, other["src"], other["pre"], M, state["w"],
                              other["w"], d_b, other["b"], dst, other["dst"], M, n_out, K, top._ld(), act, osm, other["osm"], ofmt)
                    P.pair_done = True
                    return
                _lib.call(state["fn"] + "_ex", self._h(), src, pre, M, state["w"], d_b, dst, M, n_out, K, top._ld(), act, osm, ofmt)
                return
            if M and state.get("fn") in LOWP_PAIR:
                src, pre, dst, ofmt, osm = lowp_args(M)
                _lib.call(state["fn"] + "_ex", self._h(), src, pre, M, state["w"], d_b, dst, M, n_out, K, top._ld(), act, osm, ofmt)
                return
            src = bot.dev_in(state["layout"]) if M else 0
            top.reshape(M, n_out)
            dst = top.dev_out("plain")
            if not M:
                return
            if P is not None and state["fn"] == "mnc_fc" and P.ip_prepare is not None:
                other = P.ip_prepare(M, K)
                if other is not None:
                    _lib.call("mnc_fc_pair", self._h(), src, state["w"], d_b, dst, other[0], other[1], other[2], other[3], M, n_out,
                              K, top._ld(), act)
                    P.pair_done = True
                    return
            _lib.call(state["fn"], self._h(), src, state["w"], d_b, dst, M, n_out, K, top._ld(), act)

        def prepare(M_leader, K_leader, fn_leader="mnc_fc"):
            """As the second member of a pair: (src, weights, bias, dst) of this layer's own mnc_fc call with its top made ready
            (a dict of its mnc_fc_lowp_pair arguments for the reduced-precision entry points), or None when it cannot share the
            leader's launch (other row count, other kernel, other leading dimension / activation)."""
            M = bot.shape[0]
            if M != M_leader or K != K_leader or int(np.prod(bot.shape[1:])) != K or not (bot._dev_valid or bot._host_valid or bot._sm_only):
                return None
            if "w" not in state:
                state["w"], state["layout"], state["fn"] = weights_for(bot.shape, M)
            if state["fn"] != fn_leader:
                return None
            lead_top = self.blobs[L.pair_leader.out_name or L.pair_leader.tops[0]]
            if (1 if L.pair_leader.relu else L.pair_leader.act) != act:
                return None
            if fn_leader in LOWP_PAIR:
                src, pre, dst, ofmt, osm = lowp_args(M)
                return {"src": src, "pre": pre, "w": state["w"], "b": d_b, "dst": dst, "ofmt": ofmt, "osm": osm, "ld": top._ld()}
            src = bot.dev_in(state["layout"])
            top.reshape(M, n_out)
            if top._ld() != lead_top._ld():
                return None
            return src, state["w"], d_b, top.dev_out("plain")
        if L.pair_leader is not None:
            L.ip_prepare = prepare
        return run

    def _bind_ip_group(self, L):
        members = L.group
        Ws, bs, widths = [], [], []
        for m in members:
            W, b = self._layer_weights(m)
            if m is L:
                self.params[m.name] = [_Param(W), _Param(b)]
            Ws.append(W)
            bs.append(b)
            widths.append(W.shape[0])
        K, total = Ws[0].shape[1], sum(widths)
        key = tuple(p for m in members for p in (m.param_names if m.param_names and all(m.param_names) else [m.name]))
        d_w = self._dev_param(key + ("w", "group"), lambda: self._upload(np.concatenate(Ws, 0)))
        d_b = self._dev_param(key + ("b", "group"), lambda: self._upload(np.concatenate(bs, 0)))
        bot = self.blobs[L.bottoms[0]]
        parent = Blob(self, "__group_" + L.name)
        self._hidden = getattr(self, "_hidden", [])
        self._hidden.append(parent)
        tops, off = [], 0
        for m, wd in zip(members, widths):
            t = self.blobs[m.tops[0]]
            t._view = (parent, off)
            tops.append((t, wd))
            off += wd

        def run():
            M = bot.shape[0]
            if int(np.prod(bot.shape[1:])) != K:
                raise ValueError("InnerProduct group %s: input %r does not flatten to K=%d" % (L.name, bot.shape, K))
            src = bot.dev_in("plain") if M else 0
            parent.reshape(M, total)
            dst = parent.dev_out("plain")
            for t, wd in tops:
                t.reshape(M, wd)
                t.layout = "plain"
                t._dev_valid, t._host_valid = True, False
            if M:
                _lib.call("mnc_fc", self._h(), src, d_w, d_b, dst, M, total, K, total, 0)
        return run

    def _bind_Concat(self, L, i):
        if L.msg.get1("concat_param").get1("axis", 1) != 1:
            raise NotImplementedError("Concat %s: only axis 1" % L.name)
        top = self.blobs[L.tops[0]]
        bots = [self.blobs[b] for b in L.bottoms]
        # When every bottom is produced by an InnerProduct and read by nothing else, the producers write straight
        # into column slices of the concat buffer (ld = total width) and the layer itself is a no-op.
        producers = {}
        for P in self._layers[:i]:
            if P.skip:
                continue
            for t in ([P.out_name] if P.out_name else P.tops):
                producers[t] = P
        in_place = self._fuse and all(
            producers.get(b.name) is not None and producers[b.name].type == "InnerProduct"
            and len(self._consumers.get(b.name, [])) == 1 and b.name not in self.outputs for b in bots)
        widths = [self._layer_nout(producers[b.name]) if in_place else None for b in bots]
        if in_place:
            off = 0
            for b, w in zip(bots, widths):
                b._view = (top, off)
                off += w
            total = off

            def run():
                R = bots[0].shape[0]
                top.reshape(R, total)
                top.layout = "plain"
                top._dev_valid = True
                top._host_valid = False
            # the parent must have its final shape BEFORE the producers run: hook a pre-step on the first producer
            first = min(self._layers.index(producers[b.name]) for b in bots)
            self._pre_steps = getattr(self, "_pre_steps", {})
            src_blob = self.blobs[self._layers[first].bottoms[0]]
            self._pre_steps.setdefault(first, []).append(lambda: top.reshape(src_blob.shape[0], total))
            return run

        def run():
            R = bots[0].shape[0]
            total = sum(b.shape[1] for b in bots)
            srcs = [b.dev_in("plain") for b in bots]
            top.reshape(R, total)
            dst = top.dev_out("plain")
            off = 0
            for b, s in zip(bots, srcs):
                if R:
                    _lib.call("mnc_copy2d", self._h(), dst + off * 4, total, s, b._ld(), R, b.shape[1])
                off += b.shape[1]
        return run

    @staticmethod
    def _layer_nout(L):
        return L.msg.get1("inner_product_param").get1("num_output")

    def _bind_Python(self, L, i):
        pp = L.msg.get1("python_param")
        mod = importlib.import_module(pp.get1("module"))
        layer = getattr(mod, pp.get1("layer"))()
        layer.param_str_ = pp.get1("param_str", "")
        layer.phase = self.phase
        bots = [self.blobs[b] for b in L.bottoms]
        tops = [self.blobs[t] for t in L.tops]
        self._py[L.name] = layer
        done = {}
        if self._native_py:
            native = self._native_pylayer(pp.get1("module"), pp.get1("layer"), layer, bots, tops)
            if native is not None:
                return native

        def run():
            ro = [_ReadOnlyBlob(b) for b in bots]
            if not done:
                layer.setup(ro, tops)
                done["setup"] = True
            layer.reshape(ro, tops)
            for t in tops:                      # the layer writes through top[i].data[...]
                t._host_valid, t._dev_valid = True, False
            layer.forward(ro, tops)
            for t in tops:
                t._host_valid, t._dev_valid = True, False
        return run

    def _native_pylayer(self, module, cls, layer, bots, tops):
        """Device-resident replacement for a stock Python layer, or None."""
        from mnc_config import cfg
        if (module, cls) == ("pylayer.mask_layer", "MaskLayer"):
            bot, top = bots[0], tops[0]

            def run_mask():                       # MaskLayer.forward_test: a reshape (mask_layer.py:95-102)
                R = bot.shape[0]
                src = bot.dev_in("plain")
                top.reshape(R, 1, cfg.MASK_SIZE, cfg.MASK_SIZE)
                dst = top.dev_out("plain")
                if R:
                    if bot._view is not None:
                        _lib.call("mnc_copy2d", self._h(), dst, bot.shape[1], src, bot._ld(), R, bot.shape[1])
                    else:
                        _lib.call("mnc_d2d", self._h(), dst, src, bot.count * 4)
            return run_mask
        if (module, cls) == ("pylayer.stage_bridge_layer", "StageBridgeLayer"):
            rois, bbox, probs, info = bots
            top = tops[0]

            def run_bridge():
                R, K = probs.shape[0], probs.shape[1]
                im = info._host_read()[0]
                d_rois, d_bbox, d_probs = rois.dev_in("plain"), bbox.dev_in("plain"), probs.dev_in("plain")
                top.reshape(R, 5)
                _lib.call("mnc_stage_bridge", self._h(), d_rois, d_bbox, bbox._ld(), d_probs, probs._ld(), R, K,
                          float(im[0]), float(im[1]), top.dev_out("plain"))
            return run_bridge
        if (module, cls) == ("pylayer.proposal_layer", "ProposalLayer"):
            import yaml
            from transform.anchors import generate_anchors
            prob, bbox, info = bots
            top = tops[0]
            params = yaml.safe_load(layer.param_str_) or {}
            stride = int(params["feat_stride"])
            anchors = np.ascontiguousarray(generate_anchors(), dtype=F32)
            A = anchors.shape[0]
            num = ctypes.c_int(0)

            def run_proposal():
                c = cfg[self.phase]
                post = int(c.RPN_POST_NMS_TOP_N)
                if post <= 0:
                    raise NotImplementedError("native ProposalLayer needs RPN_POST_NMS_TOP_N > 0")
                _, _, H, W = prob.shape
                im = info._host_read()[0]
                d_prob, d_bbox = prob.dev_in("plain"), bbox.dev_in("plain")
                top.reshape(post, 5)
                dst = top.dev_out("plain")
                speculate = self._speculate and self._speculated is None
                _lib.call("mnc_proposal", self._h(), d_prob, d_bbox, A, H, W, _lib.ptr(anchors), stride, float(im[0]),
                          float(im[1]), float(im[2]), int(c.RPN_PRE_NMS_TOP_N), post, float(c.RPN_NMS_THRESH),
                          float(c.RPN_MIN_SIZE), dst, None if speculate else ctypes.addressof(num))
                if speculate:                       # count stays on the device; forward() verifies it at the end
                    self._speculated = (self._running, top, post)
                else:
                    top.shape = (num.value, 5)      # same buffer, R <= post rows are valid
                top._host, top._host_valid, top._dev_valid = None, False, True
            return run_proposal
        return None

    def proposal_candidates(self):
        """(boxes [n,4], scores [n]) -- the sorted pre-NMS candidates of the last native ProposalLayer run (tests)."""
        n = ctypes.c_int(0)
        _lib.call("mnc_proposal_candidates", self._h(), None, None, 0, ctypes.addressof(n))
        boxes, scores = np.zeros((n.value, 4), F32), np.zeros(n.value, F32)
        if n.value:
            _lib.call("mnc_proposal_candidates", self._h(), _lib.ptr(boxes), _lib.ptr(scores), n.value, ctypes.addressof(n))
        return boxes, scores

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, blobs=None, start=None, end=None, _finish=True, **kwargs):
        """pycaffe's Net.forward: kwargs are input blobs; `start` / `end` name the first / last layer to run (everything
        before `start` keeps the values of the previous call -- the CFM tester re-runs only the RoI heads on the next chunk
        of proposals, lib/caffeWrapper/TesterWrapper.py:386-407); `blobs` lists extra blobs to return."""
        for name, arr in kwargs.items():
            if name not in self.inputs:
                raise KeyError("%r is not an input blob of this net (%r)" % (name, self.inputs))
            if isinstance(arr, DeviceArray):
                self.blobs[name].set_device(arr)
            else:
                self.blobs[name].set_host(arr)
        names = [L.name for L in self._layers]
        for nm in (start, end):
            if nm is not None and nm not in names:
                raise KeyError("%r is not a layer of this net" % (nm,))
        first = names.index(start) if start is not None else 0
        stop = names.index(end) + 1 if end is not None else len(self._layers)
        self._speculated = None
        self._run_layers(first, stop)
        if not _finish:
            # detect_image: the caller checks the speculative RoI count itself, with its results, and synchronises then
            return None
        if self._speculated is not None:
            # The native ProposalLayer left its RoI count on the device and the heads ran on all `post` rows (the rows are
            # independent; rows past the count are zero boxes), so the trunk -> heads hand-over needs no host round trip.
            # The count is checked here, with the outputs; in the rare case of fewer survivors the heads are run again on
            # the exact row count, which is what the reference computes.
            index, top, post = self._speculated
            self._speculated = None
            n = ctypes.c_int(0)
            _lib.call("mnc_proposal_count", self._ctx.h, ctypes.addressof(n))
            if n.value < post:
                top.shape = (n.value, 5)
                top._host, top._host_valid, top._dev_valid = None, False, True
                self._run_layers(index + 1, stop)
                _lib.call("mnc_ctx_sync", self._ctx.h)
        else:
            _lib.call("mnc_ctx_sync", self._ctx.h)
        wanted = list(self.outputs) + [b for b in (blobs or []) if b not in self.outputs]
        return _Outputs(self, [name for name in wanted if self.blobs[name]._dev_valid or self.blobs[name]._host_valid])

    def prep_image(self, im, pixel_means, factors, staged=None):
        """uint8 BGR image -> DeviceArray [len(factors),3,H',W'] = the `data` blob prep_im_for_blob / prep_im_for_blob_cfm
        (lib/utils/blob.py:36-85) build on the host, computed by mnc_prep_image (mnc_amd/prep.py).  Valid until the next call."""
        if getattr(self, "_prep", None) is None:
            from .prep import ImagePrep
            self._prep = ImagePrep(self)
        return self._prep.pyramid(im, pixel_means, factors, staged)

    def detect_tail(self, scale, im_shape):
        """The tail of im_detect (tools/demo.py:84-100) without leaving the GPU: (boxes [2R,4] in original-image pixels,
        masks [2R,1,21,21], seg scores [2R,K]) of stages 3 and 5 as DeviceArrays (valid until the next call).
        `gpu_mask_voting` consumes them in place; np.asarray() of any of them is the reference's numpy result."""
        B = self.blobs
        r1, r2 = B["rois"], B["rois_ext"]
        R1, R2 = r1.shape[0], r2.shape[0]
        m1, m2, s1, s2 = B["mask_proposal"], B["mask_proposal_ext"], B["seg_cls_prob"], B["seg_cls_prob_ext"]
        S, K = m1.shape[-1], s1.shape[1]
        n = R1 + R2
        if getattr(self, "_tail_bufs", None) is None:
            self._tail_bufs = (_DevBuf(self._ctx), _DevBuf(self._ctx), _DevBuf(self._ctx))
            self._tail_gen = 0
        self._tail_gen += 1                       # arrays of earlier images become stale (they alias these buffers)
        d_boxes = self._tail_bufs[0].ensure(max(n, 1) * 16)
        d_masks = self._tail_bufs[1].ensure(max(n, 1) * S * S * 4)
        d_scores = self._tail_bufs[2].ensure(max(n, 1) * K * 4)
        h = self._ctx.h
        _lib.call("mnc_detect_tail", h, r1.dev_in("plain") if R1 else None, R1, r2.dev_in("plain") if R2 else None, R2,
                  float(scale), int(im_shape[0]), int(im_shape[1]), d_boxes)
        for blob1, blob2, dst, width in ((m1, m2, d_masks, S * S), (s1, s2, d_scores, K)):
            if R1:
                _lib.call("mnc_copy2d", h, dst, width, blob1.dev_in("plain"), blob1._ld() if blob1._view is not None else width,
                          R1, width)
            if R2:
                _lib.call("mnc_copy2d", h, dst + R1 * width * 4, width, blob2.dev_in("plain"),
                          blob2._ld() if blob2._view is not None else width, R2, width)
        gen = (self, "_tail_gen")
        return (DeviceArray(self, d_boxes, (n, 4), self._tail_bufs, gen), DeviceArray(self, d_masks, (n, 1, S, S), self._tail_bufs, gen),
                DeviceArray(self, d_scores, (n, K), self._tail_bufs, gen))

    def vote_instances(self, boxes, masks, scores, num_classes, max_per_image, im_width, im_height, nms_thresh, iou_thresh):
        """gpu_mask_voting (lib/transform/mask_transform.py:213-286) on this net's own device-resident results (the DeviceArrays
        of detect_tail), asynchronously on the net's stream: -> a view of the net's InstanceBlock (mnc_amd/instances.py) whose
        records stay on the GPU until .fetch() / .lists() copies them down (one copy, one synchronisation) or the multi-GPU gather
        sends them.  The buffer is reused by the next image: a view that was never copied refuses to read another image's rows."""
        from .instances import InstanceBlock
        n, S = boxes.shape[0], masks.shape[-1]
        blk = getattr(self, "_inst", None)
        if blk is None or not blk.fits(num_classes, S, max_per_image, n):
            if blk is not None:
                blk.release()
            blk = self._inst = InstanceBlock(self, num_classes, S, max_per_image, n)
        blk.invalidate()
        _lib.call("mnc_vote_instances", self._ctx.h, boxes.ptr, masks.ptr, scores.ptr, n, int(num_classes), S, int(max_per_image),
                  float(nms_thresh), float(iou_thresh), int(im_height), int(im_width), blk.records_ptr, blk.rows_cap,
                  blk.counts_ptr)
        return blk.view()

    # ------------------------------------------------------------------------------------------------ one image, one launch
    def detect_image(self, im, num_classes=None, max_per_image=100, nms_thresh=None, iou_thresh=None, use_graph=True):
        """launch_image + fetch_image: one image, results on the host."""
        self.launch_image(im, num_classes, max_per_image, nms_thresh, iou_thresh, use_graph)
        return self.fetch_image()

    def launch_image(self, im, num_classes=None, max_per_image=100, nms_thresh=None, iou_thresh=None, use_graph=True):
        """First half of detect_image: returns as soon as the image's launch sequence is enqueued (fetch_image waits for it) -- a
        host that keeps two Nets in flight launches image k+1 on one before fetching image k from the other.
        tools/demo.py's per-image body -- prepare_mnc_args, net.forward, im_detect's tail, gpu_mask_voting (demo.py:54-100, 147)
        -- for ONE uint8 BGR image as one asynchronous launch sequence with a single synchronisation:
            pinned staging -> H2D -> device prep -> every layer of the prototxt -> tail -> voting -> records D2H into pinned memory
        -> (counts int32[num_classes], records float32[R, 6 + S*S]) exactly as NativeNet.forward_image / InstanceBlock.fetch return
        them.  With use_graph the sequence of an image SIZE is captured into a HIP graph the second time the size is seen
        (mnc_ctx_capture_*) and replayed afterwards: one hipGraphLaunch per image for ANY graph this engine executes (ResNet-50,
        Faster R-CNN heads excluded: the graph must end in the MNC result blobs), which is what mnc_forward_image does for the
        hand-written VGG-16 sequence.  The graph is dropped when any device buffer of the net or an arena of the context has been
        re-allocated since the capture, and an image whose RPN keeps fewer than RPN_POST_NMS_TOP_N proposals is re-run eagerly on
        the exact count (what the reference computes).  Bit-identical to the layer-by-layer path (tests/test_gpu_engine.py)."""
        from mnc_config import cfg
        from .instances import HEAD_BYTES
        im = np.ascontiguousarray(im)
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise TypeError("detect_image takes a uint8 HxWx3 image (got %s %r)" % (im.dtype, im.shape))
        K = int(num_classes or 21)
        nms_t = float(cfg.TEST.MASK_MERGE_NMS_THRESH if nms_thresh is None else nms_thresh)
        iou_t = float(cfg.TEST.MASK_MERGE_IOU_THRESH if iou_thresh is None else iou_thresh)
        H, W = im.shape[:2]
        st = self.__dict__.setdefault("_img", {"pin_in": 0, "pin_in_cap": 0, "pin_out": 0, "pin_out_cap": 0, "graphs": {}, "seen": None})
        h = self._ctx.h
        if im.nbytes > st["pin_in_cap"]:
            self._drop_image_graphs()
            if st["pin_in"]:
                _lib.call("mnc_host_free", h, st["pin_in"])
            p = ctypes.c_void_p()
            _lib.call("mnc_host_alloc", h, im.nbytes + 4096, ctypes.addressof(p))
            st["pin_in"], st["pin_in_cap"] = p.value, im.nbytes + 4096
        ctypes.memmove(st["pin_in"], im.ctypes.data, im.nbytes)          # the image is staged before anything is enqueued
        key = (H, W, K, int(max_per_image), nms_t, iou_t)
        use_graph = bool(use_graph) and not getattr(self, "_prof_level", 0)

        def body():
            """Everything asynchronous on the net's stream; returns (block view, bytes of [head | first records | count])."""
            short, long_ = min(H, W), max(H, W)
            scale = float(cfg.TEST.SCALES[0]) / float(short)
            if np.round(scale * long_) > cfg.TRAIN.MAX_SIZE:
                scale = float(cfg.TRAIN.MAX_SIZE) / float(long_)
            data = self.prep_image(im, cfg.PIXEL_MEANS, [scale], staged=st["pin_in"])
            im_info = np.array([[data.shape[2], data.shape[3], scale]], dtype=F32)
            self.blobs["data"].reshape(*data.shape)
            self.blobs["im_info"].reshape(*im_info.shape)
            self.forward(data=data, im_info=im_info, _finish=False)
            boxes, masks, scores = self.detect_tail(np.float32(scale), im.shape)
            blk = self.vote_instances(boxes, masks, scores, K, max_per_image, W, H, nms_t, iou_t)
            first = min(blk.gather_rows, blk.rows_cap)
            nbytes = HEAD_BYTES + first * blk.rec_dim * 4
            if nbytes + 64 > st["pin_out_cap"]:
                if st["pin_out"]:
                    _lib.call("mnc_host_free", h, st["pin_out"])
                p = ctypes.c_void_p()
                _lib.call("mnc_host_alloc", h, nbytes + 64, ctypes.addressof(p))
                st["pin_out"], st["pin_out_cap"] = p.value, nbytes + 64
                self._ctx.allocs += 1                                   # (a captured sequence must not outlive this buffer)
            _lib.call("mnc_d2h_async", h, st["pin_out"], blk.counts_ptr, nbytes)
            if self._speculated is not None:
                cp = ctypes.c_void_p()
                _lib.call("mnc_proposal_count_ptr", h, ctypes.addressof(cp))
                _lib.call("mnc_d2h_async", h, st["pin_out"] + nbytes, cp.value, 4)
            return blk, nbytes

        mode = "eager"
        g = st["graphs"].get(key)
        if use_graph and g is not None and g["allocs"] == self._ctx.allocs:
            try:
                _lib.call("mnc_graph_launch", h, g["graph"])
                mode = "replay"
            except _lib.MncError:
                self._drop_image_graphs()                               # an arena of the context moved: this image runs eagerly
                g = None
        elif g is not None:
            # use_graph=False / profiling on: this image bypasses its graph, which stays valid for the next one (every
            # event step of bench.py used to destroy the captured graphs of all sizes); only a moved buffer invalidates it
            if g["allocs"] != self._ctx.allocs:
                self._drop_image_graphs()
            g = None
        if mode == "eager":
            if use_graph and st["seen"] == key and key not in st.get("no_graph", ()):
                _lib.call("mnc_ctx_capture_begin", h)
                allocs0 = self._ctx.allocs
                try:
                    blk, nbytes = body()
                    gp = ctypes.c_void_p()
                    _lib.call("mnc_ctx_capture_end", h, ctypes.addressof(gp))
                except Exception:
                    # something in the sequence synchronises (a Python layer's host hop, a growing arena): the library refused that
                    # call, the capture itself is intact -- end it, throw the partial graph away, and launch this size directly
                    gp = ctypes.c_void_p()
                    try:
                        _lib.call("mnc_ctx_capture_end", h, ctypes.addressof(gp))
                    except _lib.MncError:
                        pass
                    if gp.value:
                        _lib.call("mnc_graph_destroy", gp.value)
                    st.setdefault("no_graph", set()).add(key)
                    st["seen"] = None
                    self._speculated = None
                    return self.launch_image(im, num_classes, max_per_image, nms_thresh, iou_thresh, use_graph=False)
                if allocs0 != self._ctx.allocs:                         # something was (re-)allocated while capturing
                    _lib.call("mnc_graph_destroy", gp.value)
                    blk, nbytes = body()
                else:
                    # what the pycaffe surface knows about every blob after this sequence: a replay runs no Python, so it restores
                    # this state (and drops host copies, which belong to an earlier image)
                    snap = [(b, b.shape, b.layout, b._dev_valid, b._sm, b._sm_only) for b in list(self.blobs.values()) + getattr(self, "_hidden", [])]
                    g = {"graph": gp.value, "allocs": self._ctx.allocs, "blk": blk, "nbytes": nbytes, "snap": snap,
                         "speculated": self._speculated is not None, "post": self._speculated[2] if self._speculated else 0}
                    st["graphs"][key] = g
                    _lib.call("mnc_graph_launch", h, g["graph"])
            else:
                blk, nbytes = body()
                st["seen"] = key
                g = None
        if g is not None:
            blk, nbytes, speculated, post = g["blk"], g["nbytes"], g["speculated"], g["post"]
            if mode == "replay":
                for b, shape, layout, dev_valid, sm, sm_only in g["snap"]:
                    if b._host_valid and not dev_valid and not sm_only:
                        continue                                        # an input set from the host (im_info): same values
                    b.shape, b.layout, b._dev_valid, b._sm, b._sm_only = shape, layout, dev_valid, sm, sm_only
                    b._host, b._host_valid = None, False
                blk._blk.invalidate()                                   # the buffer now holds this image
                blk = blk._blk.view()
                g["blk"] = blk
        else:
            speculated, post = self._speculated is not None, (self._speculated[2] if self._speculated else 0)
        self._speculated = None
        st["pending"] = (im, blk, nbytes, speculated, post, K, int(max_per_image), nms_t, iou_t)
        return None

    def fetch_image(self):
        """Second half of detect_image: wait for the launched image -> (counts, records)."""
        from mnc_config import cfg
        from .instances import HEAD_BYTES
        st = self.__dict__.get("_img")
        if not st or not st.get("pending"):
            raise RuntimeError("fetch_image: no image has been launched on this net")
        im, blk, nbytes, speculated, post, K, max_per_image, nms_t, iou_t = st.pop("pending")
        H, W = im.shape[:2]
        h = self._ctx.h
        _lib.call("mnc_ctx_sync", h)
        raw = (ctypes.c_char * (nbytes + 4)).from_address(st["pin_out"])
        buf = np.frombuffer(raw, dtype=np.uint8)
        if speculated and int(buf[nbytes:nbytes + 4].view(np.int32)[0]) < post:
            # fewer proposals than RPN_POST_NMS_TOP_N survived: the speculative rows were zero boxes; the reference runs the heads
            # on exactly the surviving rois -- the layer-by-layer path does that
            short, long_ = min(H, W), max(H, W)
            scale = float(cfg.TEST.SCALES[0]) / float(short)
            if np.round(scale * long_) > cfg.TRAIN.MAX_SIZE:
                scale = float(cfg.TRAIN.MAX_SIZE) / float(long_)
            data = self.prep_image(im, cfg.PIXEL_MEANS, [scale])
            self.forward(data=data, im_info=np.array([[data.shape[2], data.shape[3], scale]], dtype=F32))
            b, m, sc = self.detect_tail(np.float32(scale), im.shape)
            c2, r2 = self.vote_instances(b, m, sc, K, max_per_image, W, H, nms_t, iou_t).fetch()
            return c2.copy(), r2.copy()
        counts = buf[:K * 4].view(np.int32).copy()
        first = (nbytes - HEAD_BYTES) // (blk.rec_dim * 4)
        rec = buf[HEAD_BYTES:nbytes].view(F32).reshape(first, blk.rec_dim)
        R = int(counts[0])
        if R > first:                                                   # scores tied at the voting threshold: the rows beyond
            more = np.zeros((R - first, blk.rec_dim), F32)
            _lib.call("mnc_d2h", h, _lib.ptr(more), blk.records_ptr + first * blk.rec_dim * 4, more.nbytes)
            return counts, np.concatenate((rec.copy(), more), 0)
        return counts, rec[:R].copy()

    def _drop_image_graphs(self):
        st = self.__dict__.get("_img")
        if st:
            for g in st["graphs"].values():
                _lib.call("mnc_graph_destroy", g["graph"])
            st["graphs"].clear()
            st["seen"] = None

    def _run_layers(self, start, stop=None):
        pre = getattr(self, "_pre_steps", {})
        for L in self._layers:                     # a partial forward may have run a pair's leader without its follower
            L.pair_done = False
        for i in range(start, len(self._layers) if stop is None else stop):
            L = self._layers[i]
            if L.run is None:
                continue
            for fn in pre.get(i, ()):
                fn()
            self._running = i
            L.run()

    # ------------------------------------------------------------------------------------------------ profiling
    def profile(self, enable=True):
        """enable: False/0 off, True/1 every launch, 2 only the MFMA (>= 1 GFLOP) launches."""
        self._prof_level = int(enable)
        _lib.call("mnc_prof_enable", self._ctx.h, int(enable))
        _lib.call("mnc_prof_reset", self._ctx.h)

    def profile_enable(self, level):
        """Switch event recording on / off without touching the records collected so far (no synchronisation).  While it is on,
        detect_image launches directly: events are not part of a captured launch sequence."""
        self._prof_level = int(level)
        _lib.call("mnc_prof_enable", self._ctx.h, int(level))

    def profile_records(self):
        """[(kernel name, ms, flops, bytes)] recorded since profile(True); HIP events on the engine's stream."""
        n = ctypes.c_int(0)
        _lib.call("mnc_prof_count", self._ctx.h, ctypes.addressof(n))
        out = []
        name = ctypes.create_string_buffer(64)
        ms, fl, by = ctypes.c_float(0), ctypes.c_double(0), ctypes.c_double(0)
        for i in range(n.value):
            _lib.call("mnc_prof_get", self._ctx.h, i, ctypes.addressof(name), 64, ctypes.addressof(ms),
                      ctypes.addressof(fl), ctypes.addressof(by))
            out.append((name.value.decode(), float(ms.value), float(fl.value), float(by.value)))
        _lib.call("mnc_prof_reset", self._ctx.h)
        return out

    def sync(self):
        _lib.call("mnc_ctx_sync", self._ctx.h)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.h:
            _lib.call("mnc_ctx_sync", self._ctx.h)
            self._drop_image_graphs()
            for k in ("pin_in", "pin_out"):
                if self.__dict__.get("_img", {}).get(k):
                    _lib.call("mnc_host_free", self._ctx.h, self._img[k])
                    self._img[k] = 0
            for b in list(self.blobs.values()) + getattr(self, "_hidden", []):
                b._buf.release()
                if b._smbuf is not None:
                    b._smbuf.release()
            for t in getattr(self, "_tail_bufs", None) or ():
                t.release()
            if getattr(self, "_prep", None) is not None:
                self._prep.release()
            if getattr(self, "_inst", None) is not None:
                self._inst.release()
            self._tmp.release()
            for p in self._dev_params.values():
                self._ctx.free(p)
            self._dev_params = {}
            self._ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
