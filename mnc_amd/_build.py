"""Build mnc_amd/libmnc_hip.so for gfx950 with hipcc (in-tree, so the library travels with the repo snapshot).

    python -m mnc_amd._build [--force]

hipcc cross-compiles without a GPU.  nms.hip / mv.hip / bbox.hip / roi.hip are compiled with -ffp-contract=off: their float expressions must be evaluated
operation by operation to stay bit-exact with the reference (nms, mv) resp. the oracle's SPEC (roi: a contracted
sample coordinate moves the bilinear weights by 1 ulp, 2e-5 in the output); all four are HBM/latency-bound.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmnc_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
NO_CONTRACT = {"nms.hip", "mv.hip", "bbox.hip", "roi.hip", "proposal.hip", "prep.hip"}
# conv_wino4.hip: the transform arithmetic runs beside MFMAs as one-lane fma / add; hipcc's SLP pass would pair scalar operations of
# different window elements into v_pk_* and pay for every pair with register moves (and packed fp32 beside MFMAs costs more issue
# time than the two scalar operations: MI355X_MICROARCH.md, per-instruction constants)
EXTRA_FLAGS = {"conv_wino4.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


EXPERIMENTS = os.path.join(HERE, "..", "tools", "experiments")       # measurement-only kernels (the stream / 16x16x4 Winograd builds)


def _tuning():
    return "-DMNC_TUNING" in os.environ.get("MNC_HIPCC_EXTRA", "").split()


def sources():
    """The product library's translation units: every .hip under csrc/ (none of them is #ifdef MNC_TUNING as a whole).  A tuning
    build (MNC_HIPCC_EXTRA=-DMNC_TUNING) additionally links tools/experiments/*.hip, which csrc/conv_wino.hip then dispatches to."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _experiment_sources():
    if not _tuning() or not os.path.isdir(EXPERIMENTS):
        return []
    return sorted(f for f in os.listdir(EXPERIMENTS) if f.endswith(".hip"))


def _deps_mtime():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    paths.append(os.path.join(HERE, "..", "include", "mnc_hip.h"))
    paths.append(os.path.abspath(__file__))
    return max(os.path.getmtime(p) for p in paths)


def _base_flags():
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
            "-Wno-unused-function"] + os.environ.get("MNC_HIPCC_EXTRA", "").split()      # e.g. -DMNC_TUNING (tuning builds)


STAMP = os.path.join(OBJ, "flags.stamp")


def _flags_match():
    """Objects and library were compiled with today's command line (compiler, arch, MNC_HIPCC_EXTRA)?  mtimes alone would reuse
    objects of a tuning build (-DMNC_TUNING ...) for the product library and the other way round."""
    try:
        return open(STAMP).read() == " ".join([_hipcc()] + _base_flags())
    except OSError:
        return False


def source_hash():
    """16 hex digits over the kernel sources, the internal headers and the public header -- the identity of a build.  Profiles
    (profiles/pmc_latest.json) carry it; bench.py reports counter traffic only from a profile of the build it is running."""
    import hashlib
    h = hashlib.sha256()
    paths = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    paths.append(os.path.join(HERE, "..", "include", "mnc_hip.h"))
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_base_flags()).encode())
    return h.hexdigest()[:16]


def up_to_date():
    return os.path.isfile(LIB) and os.path.getmtime(LIB) >= _deps_mtime() and _flags_match()


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    base = [cc] + _base_flags()
    if not _flags_match():
        force = True                      # different command line: nothing compiled before may be reused

    hdr_mtime = max(os.path.getmtime(p) for p in
                    [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] +
                    [os.path.join(HERE, "..", "include", "mnc_hip.h"), os.path.abspath(__file__)])

    def one(item):
        d, src = item
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        if not force and os.path.isfile(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(d, src)), hdr_mtime):
            return obj
        cmd = base + (["-ffp-contract=off"] if src in NO_CONTRACT else []) + EXTRA_FLAGS.get(src, []) + ["-I", CSRC, "-c", os.path.join(d, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(one, [(CSRC, f) for f in sources()] + [(EXPERIMENTS, f) for f in _experiment_sources()]))
    tmp = LIB + ".tmp"
    r = subprocess.run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(" ".join(base))
    return LIB


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash())
    else:
        print(build(force="--force" in sys.argv, verbose=True))
