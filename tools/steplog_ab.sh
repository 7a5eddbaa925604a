#!/bin/bash
# per-launch log of a step with two library builds (same box): LIBA, LIBB = paths; BENCHARGS = extra bench.py arguments
set -e
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/steplog_ab; mkdir -p $o
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_product.so
trap 'cp /tmp/lib_product.so '"$pk"'/libdsmgp_hip.so' EXIT
for tag in a b; do
  if [ $tag = a ]; then cp $LIBA /tmp/l.so; else cp $LIBB /tmp/l.so; fi
  cp /tmp/l.so $pk/libdsmgp_hip.so
  DSMGP_STEPLOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCHARGS > $o/${NAME}_$tag.json 2> $o/${NAME}_steplog_$tag.txt
done
echo done
