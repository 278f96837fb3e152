#!/usr/bin/env bash
# span pass with the branch-free shuffled-input path: parity (every test that crosses the span pass in odd ways), then
# the headline and the small configs against the previous library, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_sort_stages.py -x -q -m gpu > gpurun_out/r04y_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r04y_tests.log
[ $rc -eq 0 ] || exit $rc
{
  echo "== headline: r04w (before) vs main (branch-free span rows)"
  bash tools/ab_libs.sh r04w main
  for wl in cfg3_semi_1Mx10M_24chrom cfg5_nearest_10Mx10M_24chrom cfg2_sparse_1Mx1M_1chrom; do
    echo "== $wl"
    AB_ARGS="--workload $wl" bash tools/ab_step.sh r04w main
  done
} 2>&1 | tee gpurun_out/r04y_ab.log
