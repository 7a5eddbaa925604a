#!/bin/bash
# PMC pass(es) over the depth-4 bench for one kernel-level question (run on the GPU box): tools/pmc_depth4.sh TAG "COUNTERS" [bench args]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; pmc=$2; shift 2
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out -- python3 bench.py --config dsmgp_n100k_d8_depth4 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/bench.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0)))[:8]:
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
