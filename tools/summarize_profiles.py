"""Condense gpurun_out/prof_round (tools/profile_round.sh) into the committed files under profiles/."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    """gpurun merges into gpurun_out without deleting older pulls: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
os.makedirs("profiles", exist_ok=True)
shutil.copy(newest(f"{base}/stats/runc/*_kernel_stats.csv"), f"profiles/{tag}_bench_n100k_kernel_stats.csv")
shutil.copy(f"{base}/stats/bench.log", f"profiles/{tag}_bench_n100k_rocprof_run.log")


def rows_of(d):
    f = newest(f"{base}/{d}/runc/*_counter_collection.csv")
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


out = {}
for d, name in (("pmc1", "sq_grbm"), ("pmc2", "fetch"), ("pmc3", "write_lds")):
    agg, n, dur, seen = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float), set()
    for r in rows_of(d):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
    out[name] = {"counters": {f"{k[0]}|{k[1]}": {"sum": v, "dispatches": n[k]} for k, v in agg.items()}, "kernel_seconds": dict(dur)}
g, ks = out["sq_grbm"]["counters"], out["sq_grbm"]["kernel_seconds"]
derived = {}
for name in ks:
    if f"{name}|GRBM_GUI_ACTIVE" in g and ks[name] > 0:
        gui = g[f"{name}|GRBM_GUI_ACTIVE"]["sum"] / 8.0           # summed over the 8 XCDs
        busy = g.get(f"{name}|SQ_VALU_MFMA_BUSY_CYCLES", {"sum": 0.0})["sum"] / 1024.0   # per SIMD (256 CUs x 4)
        derived[name] = {"seconds": ks[name], "clock_ghz": gui / ks[name] / 1e9, "mfma_busy_fraction": busy / gui if gui else 0.0}
out["derived"] = derived
json.dump(out, open(f"profiles/{tag}_bench_n100k_pmc_summary.json", "w"), indent=1)


def update_launches(d, counter, per_step=103):
    """tile_gemm dispatches of the last TIMED-kind step: the update launches run as tile_gemm_kernel_v2<false, 0, *> while
    per-launch timing is on (warm-up, timed steps, the breakdown step) and as <false, 2, *> in bench.py's untimed
    standalone fit/predict extra, so the last `per_step` dispatches of that name are one whole joint fit."""
    rows = [r for r in rows_of(d) if r["Counter_Name"] == counter and "tile_gemm_kernel_v2<false, 0," in r["Kernel_Name"]]
    return rows[-per_step:]


fu, wu = update_launches("pmc2", "FETCH_SIZE"), update_launches("pmc3", "WRITE_SIZE")
fetch = sum(float(r["Counter_Value"]) for r in fu) * 1024 * 2     # KB -> B; gfx950 counts 128-B requests of wide reads as 64 B
write = sum(float(r["Counter_Value"]) for r in wu) * 1024
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
import bench  # noqa: E402  (source_stamp: the tree the profile was taken on = the tree this summary is made from)
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "--", "bench.py", "include", "deepstructuredmixtures_amd/csrc"],
                           capture_output=True, text=True).stdout.strip()
    commit = commit + ("+uncommitted" if dirty else "")
except Exception:
    commit = "?"
res = {"source_stamp": bench.source_stamp(), "commit": commit,
       "kernel": "tile_gemm_kernel_v2<false, 0, false>: update launches of the last fit (test rows riding along)", "launches": len(fu),
       "fetch_bytes_total": fetch, "write_bytes_total": write, "hbm_bytes_per_launch": (fetch + write) / max(1, len(fu)),
       "note": "FETCH_SIZE (KB) x 1024 x 2 + WRITE_SIZE (KB) x 1024 over the update launches of one fit; separate --pmc passes of "
               "`python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline` (tools/profile_round.sh)"}
json.dump(res, open(f"profiles/{tag}_update_kernel_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
for k, v in derived.items():
    if v["seconds"] > 0.003:
        print(k, {a: round(b, 4) for a, b in v.items()})
