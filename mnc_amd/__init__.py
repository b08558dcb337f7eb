"""mnc_amd -- MI355X-native implementation of MNC's per-image inference hot path.

    mnc_amd/csrc   hand-written HIP kernels (gfx950) + the C ABI of include/mnc_hip.h -> libmnc_hip.so
    mnc_amd/lib    python-3 host side that keeps the reference's module paths (nms.gpu_nms, nms.mv,
                   utils.cython_bbox, pylayer.*, transform.*, utils.blob, mnc_config, caffeWrapper.TesterWrapper)
    mnc_amd/shim/caffe  the `caffe` module surface the reference's entry points use (Net / Layer / TEST / set_device)
    mnc_amd.engine device-resident executor of models/VGG16/mnc_5stage/test.prototxt

The device path has no CPU fallback: it raises if libmnc_hip.so cannot be loaded.
"""
__version__ = "0.1.0"


def install_paths():
    """Put mnc_amd/lib (reference module names: nms, pylayer, transform, utils, mnc_config, caffeWrapper) and the
    `caffe` surface on sys.path -- what the reference's tools/_init_paths.py does for lib/ and caffe-mnc/python."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "shim"), os.path.join(here, "lib")):
        if p not in sys.path:
            sys.path.insert(0, p)
