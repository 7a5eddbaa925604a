# one box: concurrent sub-contexts per GPU for shards of 8-, 4- and 2-rank jobs
for cfg in "0/8 1" "0/8 2" "0/8 3" "0/8 4" "0/4 1" "0/4 2" "0/4 3" "0/2 1" "0/2 2" "0/8 1"; do
  set -- $cfg
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sub $2 --simulate-shard $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard $1 sub $2: step', round(d['value'],4))"
done
