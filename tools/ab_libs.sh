#!/bin/bash
# Same-box A/B of library builds inside ONE gpurun call, alternating over ROUNDS rounds (default 2):
#   WHAT="h s8 s4 d4 c23" tools/ab_libs.sh TAG:lib.so[:bench args] [TAG:lib.so[:bench args] ...]
# tr = one train! iteration of the headline model, h = headline, s8 = shards 0/8 and 5/8 of an 8-rank job, s4 = shard 0/4, d4 = depth 4, d3 = depth 3, c23 = configs 2 and 3.
# One line per run on stdout; the product library is restored at the end.
set -e
cd "$GRAFT_REPO_ROOT"
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_product.so
trap 'cp /tmp/lib_product.so '"$pk"'/libdsmgp_hip.so' EXIT
o=gpurun_out/ab; mkdir -p $o
WHAT="${WHAT:-h s8 d4 c23}"
line() { tail -1 "$1" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', round(d['value'],4), 'standalone fit/predict/new rows', [round(d.get(k) or 0, 4) for k in ('standalone_fit_s', 'standalone_predict_s', 'standalone_predict_new_rows_s')], {k: round(v*1e3,2) for k, v in (d.get('device_seconds_per_step') or d.get('device_seconds_per_iteration') or {}).items() if v > 0.0005})"; }
for round in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    tag="${spec%%:*}"; rest="${spec#*:}"; lib="${rest%%:*}"; xa=""; if [ "$rest" != "$lib" ]; then xa="${rest#*:}"; fi
    cp "$lib" /tmp/lib_cur.so && cp /tmp/lib_cur.so $pk/libdsmgp_hip.so
    for w in $WHAT; do
      case $w in
        h) python bench.py --steps 3 --warmup 2 --no-cpu-baseline $xa > $o/h_${tag}_$round.json 2> $o/err.txt; line $o/h_${tag}_$round.json "$tag headline";;
        s8) for sh in 0/8 5/8; do python bench.py --steps 3 --warmup 2 --no-cpu-baseline $xa --simulate-shard $sh > $o/s_${tag}_$round.json 2> $o/err.txt; line $o/s_${tag}_$round.json "$tag shard $sh"; done;;
        s4) python bench.py --steps 3 --warmup 2 --no-cpu-baseline $xa --simulate-shard 0/4 > $o/s_${tag}_$round.json 2> $o/err.txt; line $o/s_${tag}_$round.json "$tag shard 0/4";;
        d4) python bench.py --config dsmgp_n100k_d8_depth4 --steps 3 --warmup 2 --no-cpu-baseline $xa > $o/d4_${tag}_$round.json 2> $o/err.txt; line $o/d4_${tag}_$round.json "$tag depth4";;
        d3) python bench.py --config dsmgp_n100k_d8_depth3 --steps 3 --warmup 2 --no-cpu-baseline $xa > $o/d3_${tag}_$round.json 2> $o/err.txt; line $o/d3_${tag}_$round.json "$tag depth3";;
        tr) python bench.py --mode train --steps 2 --warmup 1 $xa > $o/tr_${tag}_$round.json 2> $o/err.txt; line $o/tr_${tag}_$round.json "$tag train";;
        c23) DSMGP_RUN_ARGS="$xa" python tools/run_config3.py 2>&1 | sed "s/^/$tag /";;
      esac
    done
  done
done
echo done
