"""Phase times inside chol_diag_packed_kernel (diagnostic library): python tools/probe_diag.py [ntiles] [ld]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 144
ld = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ctx = hipabi.Context(0, diag=True)
for n in (1, 18, nt, 512, 2048):
    us, ph = ctx.probe_diag(n, ld)
    names = ["load", "blk0"] + [f"J{j}.{p}" for j in range(8) for p in ("P1", "P2")] + ["store", "inverse"]
    print(f"{n} blocks: {us:.1f} us per launch; stamps of block 0 sum {ph[:20].sum():.1f} us")
    print("   " + "  ".join(f"{a} {b:.2f}" for a, b in zip(names, ph[:20])))
    print("   wave 0 in J3.P2: trailing product %.2f  diag_block (LDS read, potrf_inv16, write) %.2f us = %d shader cycles" % tuple(ph[20:23]))
