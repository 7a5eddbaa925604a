"""Diagnostic: which ROCm runtime copies a process maps when the library's RCCL communicator comes up, by import order:
    python tools/diag_rccl_maps.py none | import | first      (no torch / torch imported after the library / torch.cuda first)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepstructuredmixtures_amd import hipabi

def maps(tag):
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("amdhip64", "rccl", "hsa-runtime"))})
    print(tag, libs, flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "first":         # the order of every torch.distributed job: torch and its device first, the library after
    import torch
    torch.cuda.init()
    x = torch.ones(4, device="cuda")
    maps("after torch.cuda.init:")
c = hipabi.Context(0)
maps("after Context:")
if mode in ("import", "init"):
    import torch
    maps("after import torch:")
    if mode == "init":
        torch.cuda.init()
        print("torch.cuda initialised", torch.cuda.is_initialized(), flush=True)
        maps("after torch.cuda.init:")
try:
    uid = hipabi.Context.comm_unique_id()
    c.comm_init(0, 1, uid)
    print("comm_init ok:", c.allgather(np.arange(3.0)))
except Exception as e:
    print("comm_init FAILED:", e)
if mode == "first":
    print("torch still works:", float((x * 2).sum().item()), flush=True)
maps("at the end:")
