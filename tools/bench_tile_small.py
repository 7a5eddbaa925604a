"""Small launches of the tile GEMM (the regime of multi-GPU shards and of the last block steps): fixed cost per
launch (K sweep at one tile per CU) and whole split-K steps as UpdateSplitter schedules them (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi
ctx = hipabi.Context(0, diag=True)
print("one tile per CU, K sweep (mode 0)")
for K in (16, 32, 64, 128, 256, 512, 720):
    tf = ctx.bench_tile(256, K, 0, 18, 10)
    print(f"  K={K:5d}: {2.0*128*128*K*256/tf/1e12*1e6:7.1f} us/launch  {tf:6.2f} TF/s", flush=True)
print("whole steps through the splitter (mode 2: split-K pieces + reduce)")
for ntiles, K in ((14, 12928), (30, 10880), (55, 9344), (117, 7808), (206, 6272), (295, 5248), (400, 4224)):
    tf = ctx.bench_tile(ntiles, K, 2, 6, 10)
    print(f"  tiles={ntiles:4d} K={K:6d}: {2.0*128*128*K*ntiles/tf/1e12*1e6:7.1f} us/step  {tf:6.2f} TF/s", flush=True)
