#!/bin/bash
# Same-box A/B of library builds inside ONE gpurun call: tools/ab_run.sh TAG:lib.so [TAG:lib.so ...]
# For every build, alternating over two rounds: headline bench, depth-4 bench, configs 2/3.  Lines land in
# gpurun_out/ab_<what>_<tag>_<round>.json (tools/ab_print.py prints them side by side).  The product library is restored.
set -e
cd "$GRAFT_REPO_ROOT"
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_product.so
trap 'cp /tmp/lib_product.so '"$pk"'/libdsmgp_hip.so' EXIT
mkdir -p gpurun_out
for round in 1 2; do
  for spec in "$@"; do
    tag="${spec%%:*}"; lib="${spec#*:}"
    cp "$lib" /tmp/lib_cur.so && cp /tmp/lib_cur.so $pk/libdsmgp_hip.so
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/ab_h_${tag}_${round}.json 2> gpurun_out/ab_h_${tag}_${round}.err
    python bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_d4_${tag}_${round}.json 2> gpurun_out/ab_d4_${tag}_${round}.err
    python tools/run_config3.py > gpurun_out/ab_c23_${tag}_${round}.log 2>&1
    echo "round $round $tag done"
  done
done
python tools/ab_print.py
grep -h "config" gpurun_out/ab_c23_*.log
