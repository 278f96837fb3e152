#!/usr/bin/env bash
# cooperative look-back round: parity of everything that sorts, then the headline and the small configs against the
# previous library, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_sort_stages.py tests/test_bucket_width.py -x -q -m gpu > gpurun_out/r04aa_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r04aa_tests.log
[ $rc -eq 0 ] || exit $rc
{
  echo "== headline: base (before) vs main (cooperative look-back round)"
  bash tools/ab_libs.sh base main
  for wl in cfg3_semi_1Mx10M_24chrom cfg5_nearest_10Mx10M_24chrom cfg2_sparse_1Mx1M_1chrom; do
    echo "== $wl"
    AB_ARGS="--workload $wl" bash tools/ab_step.sh base main
  done
} 2>&1 | tee gpurun_out/r04aa_ab.log
