#!/bin/bash
# Absolute numbers of the current build inside one gpurun call: headline, depth 4, configs 2/3 (two rounds); optional option
# overrides through bench flags in $BENCHFLAGS
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag="${1:-cur}"
for round in 1 2; do
  python bench.py --steps 3 --warmup 2 --no-cpu-baseline $BENCHFLAGS > gpurun_out/ab_h_${tag}_${round}.json 2> gpurun_out/ab_h_${tag}_${round}.err
  python bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline $BENCHFLAGS > gpurun_out/ab_d4_${tag}_${round}.json 2> gpurun_out/ab_d4_${tag}_${round}.err
  python tools/run_config3.py > gpurun_out/ab_c23_${tag}_${round}.log 2>&1
done
python tools/ab_print.py | grep "_${tag}_"
grep -h "config" gpurun_out/ab_c23_${tag}_*.log
