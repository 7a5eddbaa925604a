#!/bin/bash
# One gpurun call: parity subset on a variant library, then a same-box A/B (tools/ab_libs.sh).
#   tools/ab_call.sh TESTLIB "pytest -k expr" WHAT ROUNDS TAG:lib ...
set -e
cd "$GRAFT_REPO_ROOT"
pk=deepstructuredmixtures_amd
testlib="$1"; kexpr="$2"; what="$3"; rounds="$4"; shift 4
if [ -n "$testlib" ]; then
  cp $pk/libdsmgp_hip.so /tmp/lib_product0.so
  trap 'cp /tmp/lib_product0.so '"$pk"'/libdsmgp_hip.so' EXIT      # whatever ends this script, the product library comes back
  cp "$testlib" $pk/libdsmgp_hip.so
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
  tail -2 gpurun_out/ab_tests.log
  cp /tmp/lib_product0.so $pk/libdsmgp_hip.so
  trap - EXIT                                                         # (tools/ab_libs.sh installs its own)
fi
WHAT="$what" ROUNDS="$rounds" bash tools/ab_libs.sh "$@"
