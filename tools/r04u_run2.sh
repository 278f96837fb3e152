#!/usr/bin/env bash
# round 4: the compact-plan download of giql_hip_inner (parity, then the two e2e figures of the bench line with the
# library's own breakdown), and a finer headline A/B of the bucket-width plumbing with the phase table.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_bucket_width.py tests/test_gpu_parity.py -x -q -m gpu -k "compact or host" > gpurun_out/r04u_tests2.log 2>&1
rc=$?; tail -n 5 gpurun_out/r04u_tests2.log
[ $rc -eq 0 ] || exit $rc
GIQL_HIP_DEBUG_E2E=1 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04u_bench_e2e.json.log 2> gpurun_out/r04u_bench_e2e.err
rc=$?
grep "giql_hip_inner\]" gpurun_out/r04u_bench_e2e.err | tail -n 12
tail -n 1 gpurun_out/r04u_bench_e2e.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if k.startswith('t_e2e') and 'note' not in k}, d['ms_per_step'])"
[ $rc -eq 0 ] || exit $rc
{
  echo "== headline: previous library (r04t) vs bucket width plumbing (main), alternating, with phases"
  bash tools/ab_libs.sh r04t main
} 2>&1 | tee gpurun_out/r04u_ab2.log
