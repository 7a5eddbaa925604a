#!/bin/bash
# Same-box A/B of the number of concurrent contexts per GPU (bench.py --sub N), alternating runs: depth 4 x3, headline x2
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for round in 1 2 3; do
  for n in 1 2 3; do
    python bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline --sub $n > gpurun_out/sub_d4_${n}_${round}.json 2> gpurun_out/sub_d4_${n}_${round}.err
  done
done
for round in 1 2; do
  for n in 1 2; do
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --sub $n > gpurun_out/sub_h_${n}_${round}.json 2> gpurun_out/sub_h_${n}_${round}.err
  done
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/sub_*_*.json")):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f, round(j["value"], 4))
    except Exception as e:
        print(f, "ERR", e)
PY
