#!/usr/bin/env bash
# The N > 1 paths of bench.py on the ONE GPU this pool gives: 2 ranks over gloo (ranks share the card; the
# exchange runs through host memory -- correctness of sharding + exchange, verified against the CPU leg by
# default) and 1 rank over RCCL with the exchange forced, for every exchange implementation / gather mode.
# usage: tools/exchange_rehearsal.sh <tag>
TAG="${1:-x}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "${REPO}"
run() {
  name="$1"; shift
  echo "=== ${name}: python bench.py $*"
  timeout -k 10 280 python3 bench.py "$@" > "gpurun_out/${TAG}_${name}.json.log" 2> "gpurun_out/${TAG}_${name}.err"
  rc=$?
  echo "rc=${rc}"
  tail -n 1 "gpurun_out/${TAG}_${name}.json.log" | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print(d['n_gpus'], 'ranks', d['ms_per_step'], 'ms', d['config'].get('parallelism'), (d.get('cpu_baseline') or {}).get('parity'), d.get('exchange'))
except Exception as e:
    print('no line', e)
"
  if [ "${rc}" -ge 124 ]; then exit "${rc}"; fi
}
run g2_allgather --gpus 2 --backend gloo --steps 3 --warmup 1
run g2_p2p --gpus 2 --backend gloo --steps 3 --warmup 1 --exchange-impl p2p
run g3_root --gpus 3 --backend gloo --steps 3 --warmup 1 --gather root
run r1_allgather --force-exchange --steps 5 --warmup 2
run r1_p2p --force-exchange --steps 5 --warmup 2 --exchange-impl p2p
run r1_root --force-exchange --steps 5 --warmup 2 --gather root
run semi_g2_root --workload cfg3_semi_1Mx10M_24chrom --gpus 2 --backend gloo --steps 3 --warmup 1 --gather root
run nearest_g2_p2p --workload cfg5_nearest_10Mx10M_24chrom --gpus 2 --backend gloo --steps 3 --warmup 1 --exchange-impl p2p
run count_g3 --workload cfg3_count_1Mx10M_24chrom --gpus 3 --backend gloo --steps 3 --warmup 1
