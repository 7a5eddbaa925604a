"""Fused tile launches of two DSMGP_STEPLOG files side by side (tools/steplog_ab.sh), per block step: tasks, milliseconds and the
matrix-pipe time of the second file's executed work (every wave slot counted as holding a block: an upper bound).

    python tools/steplog_fused.py gpurun_out/steplog_ab/d4_steplog_a.txt gpurun_out/steplog_ab/d4_steplog_b.txt
"""
import re, collections, sys
def load(f):
    rows = [re.findall(r"slot (\d+) step (\d+) tasks (\d+) tiles (\d+) ms ([\d.]+)", l) for l in open(f)]
    rows = [tuple(map(float, r[0])) for r in rows if r]
    d = collections.defaultdict(list)
    for s,k,t,ti,ms in rows:
        if s == 18: d[int(k)].append((int(t), ms))
    return d
a = load(sys.argv[1]); b = load(sys.argv[2])
print("step  tasks4 ms4   | tasks8 ms8    pipe8(ms) frac")
clk=2.18e9
for k in sorted(a):
    ta = a[k][-1]; tb = b[k][-1]
    mf = tb[0]*8*(k*128/4*8 + 144)   # MFMAs per step (8 waves), upper bound (all waves active)
    valu = tb[0]*8*(16*128/64*46*4/64)  # cycles: entries per lane 32, 46 instr x 4 cycles -> in units of MFMA-equivalents /64
    pipe = (mf*64 + tb[0]*8*32*46*4)/ (1024*clk)*1e3
    print(f"{k:3d} {ta[0]:7d} {ta[1]:7.3f} | {tb[0]:7d} {tb[1]:7.3f}   {pipe:7.3f} {pipe/tb[1]:.2f}")
print(sum(a[k][-1][1] for k in a), sum(b[k][-1][1] for k in b))
