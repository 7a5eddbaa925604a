"""Cycle stamps inside the update kernel's main loop (diagnostic build): per chunk of 8 columns (32 MFMAs per wave =
2048 cycles of the f64 matrix pipe) the shader cycles of the whole chunk, of its MFMA span and of its barrier."""
import os, sys
os.environ["DSMGP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
for nt, K, mode, group in ((2048, 4096, 0, 16), (2048, 4096, 1, 16), (2048, 4096, 3, 48), (512, 4096, 0, 16), (256, 4096, 0, 16)):
    tf = ctx.bench_tile(nt, K, mode, group, 3)
    print(f"ntiles={nt} K={K} mode={mode}: {tf:6.2f} TF/s", flush=True)
