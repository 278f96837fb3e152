#!/usr/bin/env bash
# Build libgiql_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libgiql_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wall -Wno-unused-function -Wno-unused-result \
  ${GIQL_HIPCC_EXTRA:-} \
  -o "${OUT}" "${HERE}/giql_hip.hip"
echo "built ${OUT}"
