#!/bin/bash
# depth-4 experiments inside one gpurun call: concurrent sub-contexts (overlap of the latency-bound diagonal-block launches
# with the pipe-bound tile launches of another leaf group), per-step log
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for sub in 1 2 3; do
  python bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline --sub $sub > gpurun_out/d4_sub${sub}.json 2> gpurun_out/d4_sub${sub}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/d4_sub${sub}.json")); print("sub ${sub}: step", round(d["value"],4), {k: round(v*1e3,2) for k,v in d["device_seconds_per_step"].items() if v>2e-4})
PY
done
DSMGP_STEPLOG=1 python bench.py --config dsmgp_n100k_d8_depth4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/steplog_d4_r03.txt
grep "steplog" gpurun_out/steplog_d4_r03.txt | tail -60
