#!/bin/bash
# Kernel timeline of one update_cholesky! + prediction of a single GP (config 2: n = 4096) under rocprofv3 --kernel-trace:
# prints every launch of the last fit with its start offset and duration (the latency floor of the design).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_gp
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/run_config3.py > $out/run.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob('gpurun_out/trace_gp/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# config 2 runs first in run_config3.py: take the fit that ends at the 3rd mll_kernel launch (warm)
mll = [i for i, r in enumerate(rows) if "mll_kernel" in r["Kernel_Name"]]
end = mll[2]
start = max(i for i in range(end) if "gram_tile_kernel" in rows[i]["Kernel_Name"] and i < end)
while start > 0 and "gram_tile_kernel" in rows[start - 1]["Kernel_Name"]: start -= 1
t0 = int(rows[start]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void dsmgp::", "").replace("dsmgp::", "")
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    busy += e - s
    print(f"{name:42s} wgs {g:5d} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
PY
