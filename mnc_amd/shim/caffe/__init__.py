"""The slice of the `caffe` python module that MNC's inference entry points and Python layers use:

    caffe.set_mode_gpu(), caffe.set_mode_cpu(), caffe.set_device(id), caffe.TEST / caffe.TRAIN,
    caffe.Layer (base class of `type: 'Python'` layers), caffe.Net(prototxt, weights, phase)

`caffe.Net` is the device-resident MI355X executor in mnc_amd.engine; there is no CPU mode (set_mode_cpu raises)."""
TRAIN = 0
TEST = 1

_state = {"device": 0}


def set_mode_gpu():
    return None


def set_mode_cpu():
    raise RuntimeError("mnc_amd: this caffe surface is GPU-only (the MI355X HIP path); there is no CPU mode")


def set_device(device_id):
    _state["device"] = int(device_id)


def get_device():
    return _state["device"]


class Layer(object):
    """Base class of Python layers: subclasses implement setup/reshape/forward(/backward) and read
    self.param_str_ (the prototxt's python_param.param_str) and self.phase ('TEST' when str()-ed)."""
    param_str_ = ""
    phase = "TEST"

    def setup(self, bottom, top):
        pass

    def reshape(self, bottom, top):
        pass

    def forward(self, bottom, top):
        raise NotImplementedError

    def backward(self, top, propagate_down, bottom):
        raise NotImplementedError


def __getattr__(name):
    if name == "Net":           # imported lazily so that `import caffe` works in layer modules without a GPU
        from mnc_amd.engine import Net
        return Net
    raise AttributeError("module 'caffe' (mnc_amd shim) has no attribute %r" % name)
