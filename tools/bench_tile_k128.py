import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
for K in (128, 256, 512):
    for nt in (2048, 8192):
        for mode in (0, 1):
            tf = ctx.bench_tile(nt, K, mode, 16, 5)
            us = 2.0*128*128*K*nt/tf/1e12*1e6
            print(f"ntiles={nt} K={K} mode={mode}: {tf:6.2f} TF/s {us:8.1f} us/launch  {us/nt*256:6.2f} us per tile per CU", flush=True)
