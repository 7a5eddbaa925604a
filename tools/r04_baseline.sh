#!/bin/bash
# Round 4, first GPU call: today's box on the round-3 tree -- headline, depth 4 with its per-launch log, configs 2/3, and the
# two contexts per GPU on shards of 8- and 4-rank jobs, alternating.  (The run of profiles/r04_shard_schedule_ab.log also had the
# lookahead schedule in the loop: `--lookahead`, removed from the library after it.)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
o=gpurun_out/r04a
line() { tail -1 "$1" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', round(d['value'],4), {k: round(v*1e3,2) for k, v in d['device_seconds_per_step'].items() if v > 0.0005})"; }
python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $o/h.json 2> $o/h.err; line $o/h.json headline
DSMGP_STEPLOG=1 python bench.py --config dsmgp_n100k_d8_depth4 --steps 3 --warmup 2 --no-cpu-baseline > $o/d4.json 2> $o/d4_steplog.txt; line $o/d4.json depth4
python tools/run_config3.py > $o/c23.log 2>&1; cat $o/c23.log
for round in 1 2; do
  for sh in 0/8 5/8 0/4; do
    tag=${sh/\//of}
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --simulate-shard $sh > $o/s_${tag}_plain_$round.json 2> $o/s.err; line $o/s_${tag}_plain_$round.json "shard $sh plain"
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --simulate-shard $sh --sub 2 > $o/s_${tag}_sub2_$round.json 2> $o/s.err; line $o/s_${tag}_sub2_$round.json "shard $sh sub2"
  done
done
DSMGP_STEPLOG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --simulate-shard 0/8 > $o/s8_steplog.json 2> $o/s8_steplog.txt
echo done
