#!/usr/bin/env bash
# Build a variant of libgiql_hip.so into build/<name>.so with extra -D flags (A/B runs pick it up
# through GIQL_HIP_LIB).  usage: tools/build_variant.sh <name> "<-D flags>"
set -euo pipefail
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "${REPO}/build"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result \
  $2 -o "${REPO}/build/$1.so" "${REPO}/giql_amd/csrc/giql_hip.hip"
echo "built build/$1.so ($2)"
