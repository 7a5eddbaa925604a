import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
for nt, K in ((2048, 4096), (4096, 2048), (2048, 8192)):
    for mode, group in ((0, 16), (1, 16), (3, 16), (3, 48), (3, 8)):
        tf = ctx.bench_tile(nt, K, mode, group, 3)
        print(f"ntiles={nt} K={K} mode={mode} group={group}: {tf:6.2f} TF/s", flush=True)
