"""Steady-state rate of the f64-MFMA tile GEMM on uniform batches (diagnostic)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

ctx = hipabi.Context(0, diag=True)
for ntiles, K in ((256, 4096), (512, 4096), (1024, 4096), (2048, 4096), (4096, 2048), (2048, 1024), (2048, 256)):
    for mode in (0, 1):
        print(f"ntiles={ntiles:5d} K={K:5d} mode={mode}: {ctx.bench_tile(ntiles, K, mode):6.2f} TFLOP/s", flush=True)
