#!/bin/bash
# Round 4 measurement points: the profile round (PMC passes, then stats: profiles/r04_*), the headline step at the reference's
# default hyper-parameters (SURVEY 8(d) secondary point), the default bench line with its CPU baseline.
set -e
cd "$GRAFT_REPO_ROOT"
TAG=r04 bash tools/profile_round.sh
python bench.py --hyper reference-default --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_refdefault_line.json 2> gpurun_out/r04_bench_refdefault.err
tail -c 600 gpurun_out/r04_bench_refdefault_line.json; echo
python bench.py > gpurun_out/r04_bench_default_line.json 2> gpurun_out/r04_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_default_line.json').read().splitlines()[-1]); print('default', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['as_written_note'])"
