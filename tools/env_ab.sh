#!/usr/bin/env bash
# A/B of environment settings on the headline bench inside ONE gpurun call, alternating, N rounds.
# usage: tools/env_ab.sh ROUNDS "NAME1|ENV1=.. ENV2=.." "NAME2|..." ...
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
ROUNDS="$1"; shift
for rep in $(seq 1 "${ROUNDS}"); do
  for spec in "$@"; do
    name="${spec%%|*}"; envs="${spec#*|}"
    env ${envs} timeout -k 10 120 python3 "${REPO}/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s' % '$name', 'step %.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), {k: v['ms'] for k, v in d['roofline']['kernels'].items()})"
  done
done
