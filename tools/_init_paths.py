"""Put the MI355X implementation of `caffe` and of the reference's lib/ modules on sys.path (the reference's
tools/_init_paths.py does the same for caffe-mnc/python and lib/)."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)

import mnc_amd  # noqa: E402

mnc_amd.install_paths()
