#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Build the reference's OWN Cython extensions against libmnc_hip.so.

    /root/reference/lib/nms/gpu_nms.pyx  ->  oracle/_ref/ext/gpu_nms.so     (module `gpu_nms`, function gpu_nms)
    /root/reference/lib/nms/gpu_mv.pyx   ->  oracle/_ref/ext/gpu_mv.so      (module `gpu_mv`,  function mv)

cythonized as C++ (`--cplus`, what lib/setup.py:126-130, 143-147 `language='c++'` does), compiled with the reference's own
gpu_nms.hpp / gpu_mv.hpp on the include path, and linked with `-lmnc_hip` IN PLACE OF nms_kernel.cu / mv_kernel.cu -- the recipe of
INTEGRATION.md section A, executed.  The resulting modules import the mangled `_Z4_nmsPiS_PKfiifi` / `_Z3_mvPKfS0_iPKiS2_S0_iiiiiiPfPii`
from libmnc_hip.so; tests/test_gpu_nms_mv.py calls them on the GPU box against the reference-generated fixtures.

Nothing is copied into the repository: the .pyx files are read where they lie, the generated C++ goes to a temporary
directory, only the two built modules land in oracle/_ref/ext/ (git-ignored; they travel to the GPU box with the snapshot).
gpu_mv.pyx is used byte for byte.  gpu_nms.pyx needs two tokens of 2016 numpy spelled the way numpy 2 / Cython 3 still know them
(`np.int_t` -> `np.intp_t`: the dtype of argsort()'s result; `np.float thresh` -> `float thresh`); the substitution happens in
memory.  The reference's build system (lib/setup.py, nvcc) is not run.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("MNC_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref", "ext")
LIB_DIR = os.path.join(HERE, "..", "mnc_amd")
PATCHES = {"gpu_nms.pyx": [("np.ndarray[np.int_t, ndim=1]", "np.ndarray[np.intp_t, ndim=1]"), ("np.float thresh", "float thresh")],
           "gpu_mv.pyx": []}


def build(force=False):
    nms_dir = os.path.join(REF_ROOT, "lib", "nms")
    srcs = [os.path.join(nms_dir, f) for f in ("gpu_nms.pyx", "gpu_mv.pyx", "gpu_nms.hpp", "gpu_mv.hpp")]
    lib = os.path.join(LIB_DIR, "libmnc_hip.so")
    if not all(os.path.isfile(s) for s in srcs) or not os.path.isfile(lib):
        return None                       # reference not mounted (GPU box) or library not built yet: use what is prebuilt
    try:
        import Cython  # noqa: F401
        import numpy
    except ImportError:
        return None
    outs = [os.path.join(OUT_DIR, m + ".so") for m in ("gpu_nms", "gpu_mv")]
    newest = max(os.path.getmtime(p) for p in srcs + [os.path.abspath(__file__)])
    if not force and all(os.path.isfile(o) and os.path.getmtime(o) >= newest for o in outs):
        return OUT_DIR
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mnc_ref_ext_")
    try:
        for pyx, out in zip(("gpu_nms.pyx", "gpu_mv.pyx"), outs):
            with open(os.path.join(nms_dir, pyx)) as f:
                src = f.read()
            for old, new in PATCHES[pyx]:
                assert old in src, "%s: %r not found" % (pyx, old)
                src = src.replace(old, new)
            local = os.path.join(tmp, pyx)
            with open(local, "w") as f:
                f.write(src)
            cpp = local[:-4] + ".cpp"
            subprocess.run([sys.executable, "-m", "cython", "--cplus", "-3", local, "-o", cpp], check=True, capture_output=True)
            subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-w", "-I", numpy.get_include(), "-I", sysconfig.get_paths()["include"],
                            "-I", nms_dir, cpp, "-o", out, "-L", LIB_DIR, "-lmnc_hip", "-Wl,-rpath,$ORIGIN/../../../mnc_amd",
                            "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT_DIR


if __name__ == "__main__":
    d = build(force="--force" in sys.argv)
    print(d if d else "reference sources / Cython / libmnc_hip.so not available; nothing built")
