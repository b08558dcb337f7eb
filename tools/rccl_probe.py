#!/usr/bin/env python3
"""Diagnostic (GPU box): does a 1-rank RCCL communicator come up (a) through libmnc_hip.so's dlopen'd librccl without torch
in the process (system ROCm), (b) with torch imported first (torch's bundled ROCm runtime + RCCL), (c) through
torch.distributed's own nccl backend.  Prints which runtime libraries are mapped.   python tools/rccl_probe.py a|b|c"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def maps(tag):
    libs = set()
    for l in open("/proc/self/maps"):
        p = l.split()[-1]
        if any(k in p for k in ("amdhip", "hsa-runtime", "rccl")):
            libs.add(p)
    print(tag, sorted(libs), flush=True)


mode = sys.argv[1]
if mode in ("b", "c"):
    import torch
    print("torch", torch.__version__, "cuda available", torch.cuda.is_available(), flush=True)
if mode == "c":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(8, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("torch nccl all_reduce ok", t[:2].tolist(), flush=True)
    maps("c")
    dist.destroy_process_group()
    sys.exit(0)
from mnc_amd import _lib
h = ctypes.c_void_p()
_lib.call("mnc_ctx_create", ctypes.addressof(h), 0)
buf = ctypes.create_string_buffer(128)
_lib.call("mnc_comm_unique_id", ctypes.addressof(buf), 128)
maps(mode + " after unique id")
_lib.call("mnc_comm_init", h.value, ctypes.addressof(buf), 1, 0)
v = ctypes.c_int(0)
_lib.call("mnc_comm_info", h.value, None, None, ctypes.addressof(v))
print("comm ok, rccl version", v.value, flush=True)
p, q = ctypes.c_void_p(), ctypes.c_void_p()
_lib.call("mnc_dev_alloc", h.value, 4096, ctypes.addressof(p))
_lib.call("mnc_dev_alloc", h.value, 4096, ctypes.addressof(q))
_lib.call("mnc_dev_zero", h.value, p.value, 4096)
_lib.call("mnc_gather_instances", h.value, p.value, q.value, 1024)
_lib.call("mnc_ctx_sync", h.value)
print("gather ok", flush=True)
_lib.call("mnc_comm_destroy", h.value)
