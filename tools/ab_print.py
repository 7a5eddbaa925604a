"""Print the bench lines collected under gpurun_out/ab_*.json side by side (A/B runs made inside one gpurun call)."""
import glob
import json

for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    c = d["device_seconds_per_step"]
    keys = ("gram", "chol_update", "chol_fused", "chol_reduce", "chol_diag", "chol_trsm", "total_fit")
    print(f, round(d["value"], 4), "frac", round(d["roofline"]["frac"], 4),
          {k: round(v * 1e3, 2) for k, v in c.items() if k in keys}, d.get("root_mll"))
