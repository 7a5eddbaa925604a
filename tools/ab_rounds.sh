#!/bin/bash
# Same-box comparison of a previous round's tree (extracted with `git archive <commit> | tar -x -C ab/<dir>` and built in place) with
# the current one, alternating, each with its own host code and library:  tools/ab_rounds.sh ab/r3tree [rounds]
set -e
cd "$GRAFT_REPO_ROOT"
old="$1"; rounds="${2:-2}"
line() { tail -1 "$1" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', round(d['value'],4), {k: round(v*1e3,2) for k, v in d['device_seconds_per_step'].items() if v > 0.0005})"; }
o=gpurun_out/ab_rounds; mkdir -p $o
for r in $(seq 1 $rounds); do
  for tree in "$old" .; do
    tag=$(basename "$tree"); [ "$tree" = . ] && tag=current
    python $tree/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $o/h_$tag.json 2> $o/err.txt; line $o/h_$tag.json "$tag headline"
    python $tree/bench.py --config dsmgp_n100k_d8_depth4 --steps 5 --warmup 2 --no-cpu-baseline > $o/d4_$tag.json 2> $o/err.txt; line $o/d4_$tag.json "$tag depth4"
    for sh in 0/8 5/8; do python $tree/bench.py --steps 3 --warmup 2 --no-cpu-baseline --simulate-shard $sh > $o/s_$tag.json 2> $o/err.txt; line $o/s_$tag.json "$tag shard $sh"; done
    (cd $tree && python tools/run_config3.py 2>&1 | grep "^config" | sed "s/^/$tag /")
  done
done
echo done
