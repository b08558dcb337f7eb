#!/usr/bin/env python3
"""Kernels of libmnc_hip.so whose gfx950 code objects spill registers or use scratch, from the code-object notes (no GPU needed).

    python tools/spill_report.py [path/to/libmnc_hip.so]        -> one line per offending kernel; exit status 1 when there is any

Used by tests/test_abi_cpu.py (product kernels must have .vgpr_spill_count == 0)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes(lib):
    """-> {kernel name: {"vgpr_spill": n, "sgpr_spill": n, "scratch": bytes, "vgprs": n, "agprs": n, "sgprs": n, "lds": bytes}}"""
    tmp = tempfile.mkdtemp(prefix="mnc_co_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True,
                                 text=True).stdout
            cur = {}
            for line in txt.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'\"")
                if k in ("name", "symbol", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "agpr_count",
                         "sgpr_count", "group_segment_fixed_size"):
                    cur[k] = v
                if k == "wavefront_size":                        # last key of a kernel record (alphabetical order)
                    if "name" in cur and "vgpr_spill_count" in cur:
                        out[cur["name"]] = {"vgpr_spill": int(cur.get("vgpr_spill_count", 0)), "sgpr_spill": int(cur.get("sgpr_spill_count", 0)),
                                            "scratch": int(cur.get("private_segment_fixed_size", 0)), "vgprs": int(cur.get("vgpr_count", 0)),
                                            "agprs": int(cur.get("agpr_count", 0)), "sgprs": int(cur.get("sgpr_count", 0)),
                                            "lds": int(cur.get("group_segment_fixed_size", 0))}
                    cur = {}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    exe = shutil.which("c++filt")
    if not exe or not names:
        return {n: n for n in names}
    r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines())) if r.returncode == 0 else {n: n for n in names}


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mnc_amd", "libmnc_hip.so")
    notes = kernel_notes(lib)
    bad = {k: v for k, v in notes.items() if v["vgpr_spill"] or v["scratch"]}
    nice = demangle(sorted(bad))
    print("%d kernels, %d with spills / scratch" % (len(notes), len(bad)))
    for k in sorted(bad):
        v = bad[k]
        print("  %-90s vgpr_spill %d sgpr_spill %d scratch %d B (vgpr %d agpr %d)" % (nice[k][:90], v["vgpr_spill"], v["sgpr_spill"],
                                                                                     v["scratch"], v["vgprs"], v["agprs"]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
