"""Steady-state rate of the update kernel on uniform batches that fill every workgroup slot a whole number of times (no tail,
no split-K), with the shader clock held meanwhile: what fraction of the clock-scaled f64 matrix peak does the main loop
itself reach?  (diagnostic library; python tools/steady_state_tile.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

ctx = hipabi.Context(0, diag=True)
PEAK = 78.6
for ntiles, K, group in ((2048, 4096, 16), (4096, 4096, 16), (2048, 8192, 16), (4096, 2048, 32), (2048, 4096, 64)):
    for mode in (0, 1):
        ctx.bench_tile(ntiles, K, mode, group, 1)
        ctx.clock_sample_start(150.0)
        tf = ctx.bench_tile(ntiles, K, mode, group, 6)
        ghz, ms = ctx.clock_sample_read()
        print(f"ntiles={ntiles:5d} K={K:5d} group={group:3d} mode={mode}: {tf:6.2f} TFLOP/s = {tf / PEAK:.3f} of peak; clock {ghz:.3f} GHz "
              f"over {ms:.0f} ms -> {tf / (PEAK * ghz / 2.4):.3f} at the held clock", flush=True)
# the eight-wave fused tile task on the same shapes: slope of the launch time over K = its product loop alone
for ntasks, group in ((2048, 16), (4096, 16)):
    t = {K: ctx.bench_fused8(ntasks, K, group, 4) for K in (0, 2048, 4096, 8192)}
    for K0, K1 in ((2048, 4096), (4096, 8192)):
        fl = 2.0 * 128 * 128 * (K1 - K0) * ntasks
        print(f"fused8 ntasks={ntasks:5d} group={group}: K {K0} -> {K1}: {fl / (t[K1] - t[K0]) / 1e12:6.2f} TFLOP/s on the slope "
              f"(launch {t[K0] * 1e3:.3f} -> {t[K1] * 1e3:.3f} ms; K = 0: {t[0] * 1e3:.3f} ms)", flush=True)
    print(f"fused8 ntasks={ntasks:5d} whole task at K = 4096: {2.0 * 128 * 128 * 4096 * ntasks / t[4096] / 1e12:6.2f} TFLOP/s of product flops")
# lockstep: every task of a uniform launch reaches its epilogue at the same time.  Mode 6 = mode 0 with the first workgroup of
# every CU at half depth (the co-resident workgroups then run half a task apart): flops per second, counted exactly
for ntiles, K in ((2048, 4096), (4096, 4096), (2048, 8192), (3072, 6400)):
    tf0 = ctx.bench_tile(ntiles, K, 0, 16, 5)                       # TFLOP/s, every tile counted at depth K
    tf6 = ctx.bench_tile(ntiles, K, 6, 16, 5) * (ntiles - 128) / ntiles   # 256 of its tiles have half the depth
    print(f"lockstep ntiles={ntiles} K={K}: uniform {tf0:.2f} TFLOP/s; first 256 tasks halved {tf6:.2f} TFLOP/s of the flops it has", flush=True)
# marginal rate of the update kernel: launch time over K at a fixed number of tiles, and over the tiles at a fixed K
for ntiles in (2048, 4096):
    t = {K: 2.0 * 128 * 128 * K * ntiles / ctx.bench_tile(ntiles, K, 0, 16, 5) / 1e12 for K in (2048, 4096, 8192)}
    for K0, K1 in ((2048, 4096), (4096, 8192)):
        print(f"update kernel ntiles={ntiles}: K {K0} -> {K1}: {2.0 * 128 * 128 * (K1 - K0) * ntiles / (t[K1] - t[K0]) / 1e12:6.2f} TFLOP/s on the slope "
              f"(launch {t[K0] * 1e3:.3f} -> {t[K1] * 1e3:.3f} ms)", flush=True)
print("probe", ctx.probe_f64_mfma())
