#!/bin/bash
# Same-box A/B on the depth-4 config only: tools/ab_d4.sh TAG:lib.so ...   (three alternating rounds)
set -e
cd "$GRAFT_REPO_ROOT"
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_product.so
trap 'cp /tmp/lib_product.so '"$pk"'/libdsmgp_hip.so' EXIT
for round in 1 2 3; do
  for spec in "$@"; do
    tag="${spec%%:*}"; lib="${spec#*:}"
    cp "$lib" /tmp/lib_cur.so && cp /tmp/lib_cur.so $pk/libdsmgp_hip.so
    python bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_d4_${tag}_${round}.json 2> gpurun_out/ab_d4_${tag}_${round}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_d4_${tag}_${round}.json")); print("${tag} ${round}:", round(d["value"],4), {k: round(v*1e3,2) for k,v in d["device_seconds_per_step"].items() if v>2e-4})
PY
  done
done
