"""One fit of the depth-4 model launch by launch, from the kernel trace of the profile round's stats pass, against the f64-pipe work
each fused launch carries (where are the small-leaf kernels below their roofline, and is there time outside kernels?).

    python tools/fit_timeline.py gpurun_out/prof_round/depth4_stats r06          -> profiles/r06_bench_depth4_fit_timeline.json

The fit taken is a ONE-lane fit of the run (bench.py's `single_lane` extra: launches named tile_fused8_kernel<2>, one stream), whose
launches do not overlap: per launch {kernel, workgroups, start, duration}, and per fit {span, union of all launches, gap = span - union}.
Two-lane fits of the same trace are summarised beside it (span, union, sums per kernel).

Pipe-work model of a fused launch (cycles of ONE SIMD's f64 pipe per task; matrix and vector f64 instructions share that pipe --
DESIGN 4 -- and a task's waves sit on all four SIMDs of its CU alike, so CU time per task = these cycles):
  tile_fused8 task (8 waves, 16 rows x 128 columns each, two waves per SIMD):
      kernel function 2 x 32 entries per lane x 45 f64 instructions x 4 cycles            = 11,520
      block substitution 2 x 144 MFMAs x 64 cycles                                         = 18,432
      product 2 x (K / 4) x 8 MFMAs x 64 cycles                                            = 256 K
  diag_fused_reg task (4 waves, one per SIMD; nine 16x16 blocks per wave):
      kernel function 36 entries per lane x 45 x 4                                         =  6,480
      trailing products + panel solves 448 MFMAs x 64 / 4 SIMDs                            =  7,168
      eight 16x16 factorisations, ~400 dependent f64 vector instructions each, on ONE SIMD =  12,800 there (3,200 averaged)
      tile update 36 blocks x (K / 4) MFMAs x 64 / 4 SIMDs                                 = 144 K
floor of a launch = tasks / 256 CUs x cycles / clock (the clock the bench line sampled inside the steps).  Every task is counted as
full (eight row blocks of 16 data rows, eight column blocks): the floor is a little high for launches with ragged tasks.
"""
import collections
import csv
import glob
import json
import os
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round/depth4_stats"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
f = max(glob.glob(f"{base}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
line = None
for ln in reversed(open(f"{base}/bench.log").read().splitlines()):
    if ln.startswith("{"):
        line = json.loads(ln)
        break
ghz = (line or {}).get("roofline", {}).get("clock_ghz_in_steps") or 2.2

ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsmgp::", ""),
               int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Stream_Id"]))
ev.sort()
fits, cur = [], None
for e in ev:                                   # a fit = copy_vec_kernel ... mll_kernel
    if e[2].startswith("copy_vec_kernel"):
        cur = [e]
    elif cur is not None:
        cur.append(e)
        if e[2].startswith("mll_kernel"):
            fits.append(cur)
            cur = None


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)


def summary(fit):
    span = max(e[1] for e in fit) - fit[0][0]
    by = collections.defaultdict(float)
    for e in fit:
        by[e[2]] += (e[1] - e[0]) / 1e6
    return {"launches": len(fit), "streams": len({e[4] for e in fit}), "span_ms": span / 1e6, "union_ms": union([(e[0], e[1]) for e in fit]) / 1e6,
            "gap_ms": (span - union([(e[0], e[1]) for e in fit])) / 1e6, "kernel_ms_sum": {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]}}


one = [ft for ft in fits if len({e[4] for e in ft}) == 1 and any("tile_fused8_kernel" in e[2] for e in ft)]
two = [ft for ft in fits if len({e[4] for e in ft}) > 1]
out = {"trace": os.path.basename(f), "clock_ghz_in_steps": ghz, "fits_in_trace": len(fits),
       "two_lane_fits": [summary(ft) for ft in two], "one_lane_fits": [summary(ft) for ft in one]}
if one:
    fit = one[-1]
    t0 = fit[0][0]
    rows, k_f8, k_dg = [], 0, 0
    tot = {"tile_fused8": [0.0, 0.0], "diag_fused_reg": [0.0, 0.0]}
    for e in fit:
        row = {"kernel": e[2], "workgroups": e[3], "start_ms": round((e[0] - t0) / 1e6, 4), "duration_us": round((e[1] - e[0]) / 1e3, 1)}
        if e[2].startswith("tile_fused8_kernel"):
            K = 128 * k_f8
            cyc = 11520 + 18432 + 256 * K
            row.update(K=K, pipe_floor_us=round(e[3] / 256 * cyc / ghz / 1e3, 1))
            k_f8 += 1
            tot["tile_fused8"][0] += row["duration_us"]
            tot["tile_fused8"][1] += row["pipe_floor_us"]
        elif e[2].startswith("diag_fused_reg"):
            K = 128 * k_dg
            cyc = 6480 + 7168 + 3200 + 144 * K
            row.update(K=K, pipe_floor_us=round(e[3] / 256 * cyc / ghz / 1e3, 1))
            k_dg += 1
            tot["diag_fused_reg"][0] += row["duration_us"]
            tot["diag_fused_reg"][1] += row["pipe_floor_us"]
        if "pipe_floor_us" in row:
            row["floor_over_measured"] = round(row["pipe_floor_us"] / row["duration_us"], 3)
        rows.append(row)
    out["one_lane_fit_launches"] = rows
    out["one_lane_fit_totals"] = {k: {"measured_ms": round(v[0] / 1e3, 3), "pipe_floor_ms": round(v[1] / 1e3, 3),
                                      "floor_over_measured": round(v[1] / v[0], 3) if v[0] else None} for k, v in tot.items()}
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{tag}_bench_depth4_fit_timeline.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "one_lane_fit_launches"}, indent=1))
for r in out.get("one_lane_fit_launches", [])[:24]:
    print(r)
