"""Diagonal-tile (lower blocks only, tile_syrk_body) against full-tile tasks of the update kernel: time per tile.
mode 0 / 4: contiguous panels (lda = 128); mode 3 / 5: row tiles of a tall matrix (lda = 48 * 128), as in a factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
for nt, K in ((512, 512), (2048, 2048), (1024, 6144)):
    r = {}
    for mode in (0, 4, 3, 5):
        tf = ctx.bench_tile(nt // 48 * 48 if mode in (3, 5) else nt, K, mode, 48 if mode in (3, 5) else 16, 5)
        r[mode] = 2.0 * 128 * 128 * K / tf / 1e12 * 1e6 * 256
    print(f"ntiles={nt} K={K}: us per tile per CU  full {r[0]:7.2f}  sym {r[4]:7.2f} ({r[4] / r[0]:.3f})   "
          f"tall: full {r[3]:7.2f}  sym {r[5]:7.2f} ({r[5] / r[3]:.3f})", flush=True)
