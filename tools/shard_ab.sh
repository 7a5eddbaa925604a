#!/bin/bash
# Same-box A/B of the number of concurrent contexts per GPU on the shards of 2-, 4- and 8-rank jobs (one shard at a time)
cd "$GRAFT_REPO_ROOT"
for sh in 0/2 0/4 3/4 0/8 5/8; do
  for sub in 1 2 3; do
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --sub $sub --simulate-shard $sh > /tmp/sim.out 2> /tmp/sim.err
    tail -1 /tmp/sim.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard $sh sub $sub:', round(d['value'],4))"
  done
done
