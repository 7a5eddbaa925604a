#!/bin/bash
# per-rank step time of a W-rank job, simulated one shard at a time on one GPU (no exchange)
for sh in "$@"; do
  python bench.py --steps 3 --warmup 2 --no-cpu-baseline $SIMFLAGS --simulate-shard $sh > /tmp/sim.out 2> /tmp/sim.err
  grep "^# shard" /tmp/sim.err
  tail -1 /tmp/sim.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  step', round(d['value'],4), {k: round(v,4) for k,v in d['device_seconds_per_step'].items() if v > 0.001})"
done
