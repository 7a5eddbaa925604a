#!/bin/bash
# Build libdsmgp_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libdsmgp_hip.so"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result \
      -I"$here/../../include" "$here/dsmgp_hip.cpp" -o "$out" "$@"
echo "built $out"
