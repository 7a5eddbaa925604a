#!/bin/bash
# Round 4: the diagonal block inside the update launch (DSMGP_OPT_DIAG_IN_UPDATE) -- parity test, then same-box A/B against the
# library built before the change (deepstructuredmixtures_amd/libdsmgp_hip_prev.so), alternating: headline, 8-rank shards, config 2/3, depth 4.
set -e
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04b; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diagonal_block_inside or fused_steps_agree or first_bad_minor or single_gp_vs or refit or config1" > $o/pytest.log 2>&1 || { tail -30 $o/pytest.log; exit 1; }
tail -3 $o/pytest.log
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_new.so
line() { tail -1 "$1" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', round(d['value'],4), {k: round(v*1e3,2) for k, v in d['device_seconds_per_step'].items() if v > 0.0005})"; }
trap 'cp /tmp/lib_new.so '"$pk"'/libdsmgp_hip.so' EXIT
for round in 1 2; do
  for tag in prev new; do
    if [ $tag = prev ]; then cp deepstructuredmixtures_amd/libdsmgp_hip_prev.so $pk/libdsmgp_hip.so; else cp /tmp/lib_new.so $pk/libdsmgp_hip.so; fi
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $o/h_${tag}_$round.json 2> $o/err.txt; line $o/h_${tag}_$round.json "headline $tag"
    for sh in 0/8 5/8 0/4; do
      python bench.py --steps 3 --warmup 2 --no-cpu-baseline --simulate-shard $sh > $o/s_${tag}_$round.json 2> $o/err.txt; line $o/s_${tag}_$round.json "shard $sh $tag"
    done
    python bench.py --config dsmgp_n100k_d8_depth4 --steps 3 --warmup 2 --no-cpu-baseline > $o/d4_${tag}_$round.json 2> $o/err.txt; line $o/d4_${tag}_$round.json "depth4 $tag"
    python tools/run_config3.py 2>&1 | sed "s/^/$tag /"
  done
done
echo done
