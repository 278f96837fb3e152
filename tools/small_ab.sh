#!/usr/bin/env bash
# A/B of environment switches on the small configs, alternating runs.
# usage: tools/small_ab.sh "<env settings, ';'-separated, '-' = none>" "<workloads>"
IFS=';' read -ra ENVS <<< "${1:--}"
WLS="${2:-cfg2_sparse_1Mx1M_1chrom}"
for rep in 1 2 3; do
for wl in $WLS; do
  for e in "${ENVS[@]}"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    env $ee timeout -k 10 200 python3 bench.py --workload "$wl" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$wl [$e]', round(d['ms_per_step'],4), d.get('parity'))"
  done
done
done
