for cfg in "4 1" "8 1" "4 2" "8 2" "6 1" "4 1"; do
  set -- $cfg
  DSMGP_TAIL_SPLIT=$1 DSMGP_TAIL_ROUNDS=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('split $1 rounds $2: step', round(d['value'],4), 'update', round(d['device_seconds_per_step']['chol_update'],4), 'achieved', round(d['roofline']['achieved'],2))"
done
