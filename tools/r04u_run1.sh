#!/usr/bin/env bash
# round 4, bucket width by density: parity tests of the narrow forms, the headline A/B against the previous library,
# the host expansion probe.  usage (on the GPU box): bash tools/r04u_run1.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_bucket_width.py tests/test_sort_stages.py tests/test_index_gpu.py -x -q -m gpu > gpurun_out/r04u_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/r04u_tests.log
[ $rc -eq 0 ] || exit $rc
g++ -O3 -march=native -pthread -o /tmp/host_expand_probe tools/probes/host_expand_probe.cpp && timeout -k 5 120 /tmp/host_expand_probe | tee gpurun_out/r04u_host_expand_probe.log
{
  echo "== headline: previous library (r04t) vs bucket width plumbing (main), alternating"
  bash tools/ab_step.sh r04t main
} 2>&1 | tee gpurun_out/r04u_ab.log
