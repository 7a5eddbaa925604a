"""Diagnostic: which ROCm runtime copies a process maps when the library's RCCL communicator comes up after torch was imported."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepstructuredmixtures_amd import hipabi

def maps(tag):
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("amdhip64", "rccl", "hsa-runtime"))})
    print(tag, libs, flush=True)

c = hipabi.Context(0)
maps("after Context:")
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
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
maps("at the end:")
