#!/usr/bin/env bash
# A/B of library variants built by tools/build_variant.sh: runs bench.py with each, twice,
# alternating, and prints step ms + the phase table.  usage: tools/ab_libs.sh name1 name2 ...
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for rep in 1 2; do
  for n in "$@"; do
    lib="${REPO}/build/${n}.so"; [ "$n" = main ] && lib="${REPO}/giql_amd/libgiql_hip.so"; GIQL_HIP_LIB="$lib" timeout -k 10 120 python3 "${REPO}/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$n', 'step %.3f ms' % d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})"
  done
done
