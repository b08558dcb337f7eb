#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Build oracle/_ref/libmnc_ref.so from the reference's own sources.

Compiles, *where they lie* (nothing is copied into this repository):
    /root/reference/lib/nms/nms_kernel.cu   -> exports  _nms(...)   (lib/nms/gpu_nms.hpp:1-2)
    /root/reference/lib/nms/mv_kernel.cu    -> exports  _mv(...)    (lib/nms/gpu_mv.hpp:1-4)
for the host CPU with g++ through oracle/cuda_on_cpu.h.  The only source transformation is the
CUDA launch syntax, which g++ cannot parse:  `k<<<g, b>>>(args);`  ->  `MNC_CPU_LAUNCH(mode, k, g, b, args);`
It is applied in memory and the result is piped to the compiler's stdin.

The reference's own build system (lib/setup.py + nvcc) is NOT run.  Output goes only to
oracle/_ref/ (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("MNC_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "libmnc_ref.so")

LAUNCH_RE = re.compile(r"(\w+)\s*<<<\s*(.*?)\s*>>>\s*\((.*?)\)\s*;", re.S)


def _split_top_level(s):
    """Split 'a(b,c), d' at the top-level comma -> ['a(b,c)', 'd']."""
    depth, parts, cur = 0, [], []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur).strip())
    return parts


def rewrite_launches(src):
    uses_barrier = set()
    # a kernel needs fiber mode iff its body calls __syncthreads()
    for m in re.finditer(r"__global__\s+void\s+(\w+)\s*\(", src):
        name = m.group(1)
        start = src.index("{", m.end())
        depth, i = 0, start
        while True:
            if src[i] == "{":
                depth += 1
            elif src[i] == "}":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        if "__syncthreads" in src[start:i]:
            uses_barrier.add(name)

    def repl(m):
        kernel, cfg, args = m.group(1), m.group(2), m.group(3)
        g, b = _split_top_level(cfg)[:2]
        mode = "fiber" if kernel in uses_barrier else "direct"
        return "MNC_CPU_LAUNCH(%s, %s, %s, %s, %s);" % (mode, kernel, g, b, " ".join(args.split()))

    out, n = LAUNCH_RE.subn(repl, src)
    return out, n


def compile_cu(path, obj):
    with open(path, "r") as f:
        src, n = rewrite_launches(f.read())
    assert n > 0, "no kernel launch found in %s" % path
    cmd = ["g++", "-x", "c++", "-std=c++14", "-O2", "-fPIC", "-fopenmp", "-ffp-contract=off",
           "-include", os.path.join(HERE, "cuda_on_cpu.h"),
           "-I", os.path.dirname(path), "-I", HERE, "-w", "-c", "-", "-o", obj]
    subprocess.run(cmd, input=src.encode(), check=True)


def build(force=False):
    nms_cu = os.path.join(REF_ROOT, "lib", "nms", "nms_kernel.cu")
    mv_cu = os.path.join(REF_ROOT, "lib", "nms", "mv_kernel.cu")
    if not (os.path.isfile(nms_cu) and os.path.isfile(mv_cu)):
        return None  # reference not mounted (e.g. on the GPU box): use the prebuilt .so if present
    srcs = [nms_cu, mv_cu, os.path.join(HERE, "cuda_on_cpu.cpp"), os.path.join(HERE, "cuda_on_cpu.h"),
            os.path.abspath(__file__)]
    if (not force and os.path.isfile(OUT_SO)
            and os.path.getmtime(OUT_SO) >= max(os.path.getmtime(s) for s in srcs)):
        return OUT_SO
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    for cu in (nms_cu, mv_cu):
        obj = os.path.join(OUT_DIR, os.path.basename(cu) + ".o")
        compile_cu(cu, obj)
        objs.append(obj)
    rt = os.path.join(OUT_DIR, "cuda_on_cpu.o")
    subprocess.run(["g++", "-std=c++14", "-O2", "-fPIC", "-fopenmp", "-c",
                    os.path.join(HERE, "cuda_on_cpu.cpp"), "-o", rt], check=True)
    subprocess.run(["g++", "-shared", "-fopenmp", "-o", OUT_SO] + objs + [rt], check=True)
    for o in objs + [rt]:
        os.remove(o)
    return OUT_SO


if __name__ == "__main__":
    so = build(force="--force" in sys.argv)
    print(so if so else "reference sources not found under %s; nothing built" % REF_ROOT)
