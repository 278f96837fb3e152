#!/usr/bin/env bash
# A/B of library variants (tools/build_variant.sh) on any bench workload: ms per step only, alternating, twice.
# usage: AB_ARGS="--workload cfg5_nearest_10Mx10M_24chrom" tools/ab_step.sh name1 name2 ...   ("main" = giql_amd/libgiql_hip.so)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for rep in 1 2; do
  for n in "$@"; do
    lib="${REPO}/build/${n}.so"; [ "$n" = main ] && lib="${REPO}/giql_amd/libgiql_hip.so"
    GIQL_HIP_LIB="$lib" timeout -k 10 180 python3 "${REPO}/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$n', 'step %.4f ms  median %.4f' % (d['ms_per_step'], d['ms_per_step_median']))"
  done
done
