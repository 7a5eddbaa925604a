"""Launch time of the tile GEMM vs number of tiles: raw launch vs the split scheduler (tail pieces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
for K in (2048, 6144):
    for n in (260, 516, 772, 1028, 1540, 2048, 2646, 4096, 6000):
        r = []
        for mode in (0, 2):
            tf = ctx.bench_tile(n, K, mode, 16, 4)
            r.append(2.0 * 128 * 128 * K * n / tf / 1e9)
        print(f"K={K} ntiles={n:5d}  raw {r[0]:7.3f} ms   scheduled {r[1]:7.3f} ms  ({100*(r[1]/r[0]-1):+.1f} %)", flush=True)
