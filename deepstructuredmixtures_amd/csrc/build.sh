#!/bin/bash
# Build libdsmgp_hip.so for gfx950 in-tree (cross-compiles without a GPU).
#   build.sh        the product library
#   build.sh diag   libdsmgp_hip_diag.so: the same sources with -DDSMGP_DIAG (cycle stamps, micro-benchmarks,
#                   scheduling knobs from the environment; tools/ only, never loaded by the package)
#   OUT=path build.sh [-D...]   a variant build for same-box A/B runs (tools/ab_libs.sh), written to `path`
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libdsmgp_hip.so"
extra=()
if [ "${1:-}" = "diag" ]; then
    shift
    out="$here/../libdsmgp_hip_diag.so"
    extra=(-DDSMGP_DIAG)
fi
out="${OUT:-$out}"
# host_tree.cpp (tree builder, main-leaf search) is plain C++: no offload pass, and no fused multiply-add -- the builder
# promises NumPy's floating-point results bit for bit
obj="$(mktemp -d)"
trap 'rm -rf "$obj"' EXIT
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -fPIC -Wall -ffp-contract=off -pthread -c "$here/host_tree.cpp" -o "$obj/host_tree.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -pthread \
      -I"$here/../../include" "${extra[@]}" -c "$here/dsmgp_hip.cpp" -o "$obj/dsmgp_hip.o" "$@"
hipcc --offload-arch=gfx950 -fPIC -shared -pthread "$obj/dsmgp_hip.o" "$obj/host_tree.o" -ldl -o "$out"
echo "built $out"
