#!/bin/bash
# predict(model, x) on new rows with variant libraries, same box: tools/ab_newrows.sh config TAG:lib ...
set -e
cd "$GRAFT_REPO_ROOT"
pk=deepstructuredmixtures_amd
cp $pk/libdsmgp_hip.so /tmp/lib_product.so
trap 'cp /tmp/lib_product.so '"$pk"'/libdsmgp_hip.so' EXIT
cfg="$1"; shift
for spec in "$@"; do
  tag="${spec%%:*}"; lib="${spec#*:}"
  cp "$lib" /tmp/lib_cur.so && cp /tmp/lib_cur.so $pk/libdsmgp_hip.so
  python tools/time_predict_new_rows.py "$cfg" 2>&1 | grep "predict on new rows" | tail -4 | sed "s/^/$tag /"
done
