"""When does each leaf lane finish?  DSMGP_STEPLOG=1 python bench.py ... 2> log; python tools/lane_timeline.py log
The step log carries, per timed launch, its start relative to the start of the call: the update launches (slot 1) of one block
step appear once per lane, in lane order; the last launch of each lane gives the time that lane's chain ends."""
import re
import sys

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"steplog slot (\d+) step (\d+) tasks (\d+) tiles (\d+) ms ([\d.]+) at ([-\d.]+)", ln)
    if m:
        rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6))))
# split into calls: the start offsets restart
calls, cur, last = [], [], -1.0
for r in rows:
    if r[5] < last - 50.0 and cur:
        calls.append(cur)
        cur = []
    cur.append(r)
    last = max(last, r[5]) if cur and len(cur) > 1 else r[5]
calls.append(cur)
for ci, call in enumerate(calls[-int(sys.argv[2]) if len(sys.argv) > 2 else -3:]):
    upd = [r for r in call if r[0] == 1]
    # lanes: within a step the launches are logged lane after lane
    lanes = {}
    seen = {}
    for r in upd:
        q = seen.get(r[1], 0)
        seen[r[1]] = q + 1
        lanes.setdefault(q, []).append(r)
    print(f"call {ci}: {len(upd)} update launches")
    for q, rs in sorted(lanes.items()):
        end = max(r[5] + r[4] for r in rs)
        print(f"  lane {q}: {len(rs)} launches, steps {rs[0][1]}..{rs[-1][1]}, sum {sum(r[4] for r in rs):.2f} ms, last launch ends at {end:.2f} ms")
