"""Tile GEMM rate vs the leading dimension of the A operand (contiguous panels vs row tiles of a narrow matrix)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
K = 4096
print("contiguous panels (lda=128):", round(ctx.bench_tile(1536, K, 0, 16, 3), 2))
for g in (2, 4, 5, 6, 8, 11, 16, 48):
    print(f"row tiles of a {g*128} x K matrix (lda={g*128}):", round(ctx.bench_tile(1536 // g * g, K, 3, g, 3), 2), flush=True)
