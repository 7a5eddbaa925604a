#!/bin/bash
# Build libdsmgp_hip.so for gfx950 in-tree (cross-compiles without a GPU).
#   build.sh        the product library
#   build.sh diag   libdsmgp_hip_diag.so: the same sources with -DDSMGP_DIAG (cycle stamps, micro-benchmarks,
#                   scheduling knobs from the environment; tools/ only, never loaded by the package)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libdsmgp_hip.so"
extra=()
if [ "${1:-}" = "diag" ]; then
    shift
    out="$here/../libdsmgp_hip_diag.so"
    extra=(-DDSMGP_DIAG)
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -pthread -ldl \
      -I"$here/../../include" "${extra[@]}" "$here/dsmgp_hip.cpp" -o "$out" "$@"
echo "built $out"
