# one box: default vs two concurrent sub-contexts, full job and an 8-rank shard
for sub in 1 2 1 2; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sub $sub 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sub $sub full: step', round(d['value'],4))"
done
for sub in 1 2 3; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sub $sub --simulate-shard 0/8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sub $sub shard 0/8: step', round(d['value'],4))"
done
