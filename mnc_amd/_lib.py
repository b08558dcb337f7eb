"""ctypes binding of libmnc_hip.so.  The prototypes are parsed from include/mnc_hip.h, so the Python side can never
drift from the C ABI.  There is NO fallback: if the library cannot be loaded, importing the device path raises."""
import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "mnc_hip.h")
LIB_PATH = os.environ.get("MNC_LIB_PATH") or os.path.join(HERE, "libmnc_hip.so")    # MNC_LIB_PATH: an experiment build (tools/probes)

MNC_OK = 0


class MncError(RuntimeError):
    def __init__(self, code, func, msg):
        RuntimeError.__init__(self, "%s failed (status %d): %s" % (func, code, msg))
        self.code = code


_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "size_t": ctypes.c_size_t, "double": ctypes.c_double,
    "void": None,
}


def _map_type(t):
    t = t.replace("const ", "").strip()
    if t.endswith("*"):
        return ctypes.c_void_p        # every pointer argument is passed as a raw address (host or device)
    return _CTYPES[t]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every MNC_API declaration."""
    with open(path) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"MNC_API\s+([\w\s\*]+?)\s*\b(\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)(\w+)$", a)
                argtypes.append(_map_type(mm.group(1)))
                argnames.append(mm.group(2))
        restype = ctypes.c_char_p if "char" in ret else _map_type(ret)
        decls[name] = (restype, argtypes, argnames)
    return decls


_lib = None
_decls = None


def load():
    """Load (building first if the .so is missing and hipcc is available).  Raises if that is impossible."""
    global _lib, _decls
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        from . import _build
        _build.build()
    lib = ctypes.CDLL(LIB_PATH)
    _decls = parse_header()
    for name, (restype, argtypes, _) in _decls.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-status entry point; raise MncError with mnc_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != MNC_OK:
        raise MncError(rc, name, lib.mnc_last_error().decode("utf-8", "replace"))
    return rc


def ptr(a):
    """Address of a C-contiguous numpy array (kept alive by the caller), or pass-through for ints/None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return a.ctypes.data
    return a


def device_count():
    n = ctypes.c_int(0)
    call("mnc_device_count", ctypes.addressof(n))
    return n.value


def device_mem_info(device_id=0):
    """-> (free bytes, total bytes) of a device."""
    fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
    call("mnc_device_mem_info", int(device_id), ctypes.addressof(fr), ctypes.addressof(tot))
    return fr.value, tot.value
