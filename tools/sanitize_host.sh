#!/bin/bash
# Host-only code of the library (csrc/host_tree.cpp: tree builder, leaf means, main-leaf search with its worker threads)
# under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU (GPU sanitizers are not available on the pool).
# Builds tools/micro/host_sanitize.cpp, which includes the translation unit itself, into /tmp and runs eight tree shapes.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
cxx=/opt/rocm/lib/llvm/bin/clang++
out="$(mktemp -d)"
trap 'rm -rf "$out"' EXIT
cd "$here/micro"
$cxx -O1 -g -std=c++17 -ffp-contract=off -pthread -fsanitize=address,undefined -fno-omit-frame-pointer host_sanitize.cpp -o "$out/asan"
"$out/asan"
$cxx -O1 -g -std=c++17 -ffp-contract=off -pthread -fsanitize=thread host_sanitize.cpp -o "$out/tsan"
"$out/tsan"
echo "host code: clean under ASan+UBSan and TSan"
