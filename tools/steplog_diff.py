"""Per-step comparison of two DSMGP_STEPLOG files (tools/steplog_ab.sh): update launches (slot 1) and the chain (slots 2, 3)."""
import re, sys, collections

def load(f):
    rows = [re.findall(r"slot (\d+) step (\d+) tasks (\d+) tiles (\d+) ms ([\d.]+)", l) for l in open(f)]
    rows = [tuple(map(float, r[0])) for r in rows if r]
    idx = [i for i, r in enumerate(rows) if r[0] == 1]
    m = min(rows[i][1] for i in idx)
    last = [i for i in idx if rows[i][1] == m][-1]
    d = collections.defaultdict(dict)
    for slot, k, tasks, tiles, ms in rows[last:]:
        d[int(k)][int(slot)] = (int(tasks), int(tiles), ms)
    return d

a, b = load(sys.argv[1]), load(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ta = tb = 0.0
print("step | A: update(tasks, tiles, ms) diag ms trsm ms | B: ... | update delta us")
for k in sorted(a):
    if 1 not in a[k] or 1 not in b.get(k, {}):
        continue
    ta += a[k][1][2]
    tb += b[k][1][2]
    if k % every == 1 or k > max(a) - 6:
        fa = lambda d: (d.get(1), round(d.get(2, (0, 0, 0))[2] * 1e3), round(d.get(3, (0, 0, 0))[2] * 1e3))
        print(k, fa(a[k]), fa(b[k]), f"{(b[k][1][2] - a[k][1][2]) * 1e3:+.0f}")
print("update totals ms:", round(ta, 2), round(tb, 2))
