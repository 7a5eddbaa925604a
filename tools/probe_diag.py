"""Phase times inside chol_diag_kernel (diagnostic): python tools/probe_diag.py [ntiles] [ld]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 144
ld = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ctx = hipabi.Context(0, diag=True)
for n in (1, 18, nt):
    us, ph = ctx.probe_diag(n, ld)
    names = ["load"] + [f"J{j}.{p}" for j in range(8) for p in ("P1", "P2")] + ["store", "z"]
    print(f"{n} blocks: {us:.1f} us per launch; stamps of block 0 sum {ph[:19].sum():.1f} us")
    print("   " + "  ".join(f"{a} {b:.1f}" for a, b in zip(names, ph[:19])))
    print("   wave 0 in J3.P2: product %.2f  LDS read %.2f  potrf_inv16 %.2f  LDS write %.2f" % tuple(ph[19:]))
