"""How launch time of the tile GEMM depends on the number of tiles (dispatch quantisation), raw vs split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0)
for K in (2048, 6144):
    for n in (64, 128, 200, 256, 260, 384, 516, 768, 772, 900, 1028, 1540, 2052):
        r = []
        for mode in (0, 2):
            tf = ctx.bench_tile(n, K, mode, 16, 5)
            r.append(2.0 * 128 * 128 * K * n / tf / 1e9)
        print(f"K={K} ntiles={n:5d}  raw {r[0]:7.3f} ms   scheduled {r[1]:7.3f} ms", flush=True)
