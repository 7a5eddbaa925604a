# usage: sweep_env.sh VAR v1 v2 ...   -- one bench line per value, same box
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['device_seconds_per_step']; print('$var=$v: step', round(d['value'],4), 'update', round(c['chol_update'],4), 'trsm', round(c['chol_trsm'],4), 'diag', round(c['chol_diag'],4), 'solve', round(c['solve'],4))"
done
