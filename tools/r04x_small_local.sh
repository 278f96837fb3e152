#!/usr/bin/env bash
# would the small INNER configs gain from the three-stage sort + the join in the bucket stage?  (forced by
# GIQL_HIP_LOCAL_MIN_ROWS, which also lifts the density gates)
for rep in 1 2; do
for wl in cfg2_sparse_1Mx1M_1chrom cfg2_dense_1Mx1M_1chrom cfg4_small_1Mx10M_24chrom; do
  for lm in default 500000; do
    if [ $lm = default ]; then unset GIQL_HIP_LOCAL_MIN_ROWS; else export GIQL_HIP_LOCAL_MIN_ROWS=$lm; fi
    timeout -k 10 200 python3 bench.py --workload $wl --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s local_min_rows %-8s step %.4f ms' % ('$wl', '$lm', d['ms_per_step']), d['config'].get('join_form'), d['config'].get('pairs_written_by','')[:30], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})"
  done
done
done
