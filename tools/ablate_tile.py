"""Ablation of the pipelined tile GEMM (micro-benchmark, L2-resident operands): which phase costs MFMA issue slots."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); from deepstructuredmixtures_amd import hipabi; c = hipabi.Context(0); "
        "print(' '.join('%%6.2f' %% c.bench_tile(n, 4096, 1) for n in (256, 512, 2048)))" % root)
names = {2: "full kernel", 108: "gload kept, no swrite", 101: "no gload/swrite", 102: "no barrier", 104: "no frag reads", 103: "no gload, no barrier",
         105: "no gload, no frags", 106: "no barrier, no frags", 107: "MFMA only"}
print("variant                      TF/s @256   @512  @2048 tiles (K=4096, shared operands)")
for v, name in names.items():
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DSMGP_TILE_V=str(v)), capture_output=True, text=True)
    print(f"{name:28s} {out.stdout.strip()} {out.stderr.strip()[-200:]}")
