"""Per-launch efficiency of the factorisation steps (diagnostic).

    DSMGP_STEPLOG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/steplog.txt
    python tools/steplog.py gpurun_out/steplog.txt

Uses the records of the last logged fit (the level-2 step): for every update launch the executed tile flops
(whole tiles x 2*128*128*K) over its time (split-K reduce included), next to the tile count in units of 256.
"""
import sys, re, collections

rows = [re.findall(r"slot (\d+) step (\d+) tasks (\d+) tiles (\d+) ms ([\d.]+)", l) for l in open(sys.argv[1])]
rows = [tuple(map(float, r[0])) for r in rows if r]
# split into fits: a fit starts when slot-1 step index decreases to its minimum again
fits, cur, last = [], [], -1
for r in rows:
    if r[0] == 1 and r[1] < last:
        fits.append(cur); cur = []
    if r[0] == 1:
        last = r[1]
    cur.append(r)
fits.append(cur)
# the fit with the most update launches among the fully logged ones (a PREFIX phase logs a short second sequence)
fit = max([f for f in fits if any(r[0] == 2 for r in f)], key=lambda f: sum(1 for r in f if r[0] == 1))
tot = collections.defaultdict(float)
print("step  tiles  tasks  rounds   ms     TF/s(exec)")
fl_all = t_all = 0.0
for slot, k, tasks, tiles, ms in fit:
    tot[int(slot)] += ms
    if slot == 1:
        fl = tiles * 2.0 * 128 * 128 * k * 128
        fl_all += fl; t_all += ms
        if int(k) % 4 == 1 or k > 100:
            print(f"{int(k):4d} {int(tiles):6d} {int(tasks):6d} {tiles/256:7.2f} {ms:7.3f} {fl/ms/1e9:8.1f}")
print("totals ms by slot (1 update, 2 diag, 3 panel solve):", {k: round(v, 2) for k, v in tot.items()})
print(f"update: executed {fl_all/1e12:.2f} TFLOP in {t_all:.1f} ms = {fl_all/t_all/1e9:.1f} TF/s")
