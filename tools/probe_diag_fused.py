"""The fused steps' diagonal-block task (diag_fused_reg_kernel) alone, by how many tasks share a CU (diagnostic library):
256 / 512 / 768 blocks = one / two / three workgroups on every CU -- the latency of one task, and what co-residents cost it.

    python tools/probe_diag_fused.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

ctx = hipabi.Context(0, diag=True)
for K in (0, 128, 256, 512):
    base = None
    for nt in (64, 256, 512, 768, 1024, 1536, 2304, 3072, 6144, 18432):
        us = ctx.probe_diag_fused(nt, K, 10)
        rounds = nt / 768.0
        base = base or us
        print(f"K={K:4d} blocks={nt:6d} ({nt / 256:5.1f} per CU): {us:8.1f} us per launch, {us / max(1.0, rounds):7.1f} us per round of 768, "
              f"{us / nt * 256:6.2f} us per block and CU", flush=True)
print("probe", ctx.probe_f64_mfma_detail(8))
