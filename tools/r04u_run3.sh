#!/usr/bin/env bash
# round 4: narrow buckets at scale (35M x 350M parity; 40M x 400M timed with and without them), the compact-plan e2e
# with non-temporal expansion stores, the headline A/B against the previous library.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_bucket_width.py -x -q -m gpu -k "compact or density or every_operator" > gpurun_out/r04u_tests3.log 2>&1
rc=$?; tail -n 3 gpurun_out/r04u_tests3.log
[ $rc -eq 0 ] || exit $rc
{
  echo "== headline: previous library (r04t) vs main, alternating, with phases"
  bash tools/ab_libs.sh r04t main
} 2>&1 | tee gpurun_out/r04u_ab3.log
for nt in 1 0; do
  GIQL_HIP_E2E_NT=$nt GIQL_HIP_DEBUG_E2E=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04u_bench_e2e_nt$nt.json.log 2> gpurun_out/r04u_bench_e2e_nt$nt.err || exit 1
  echo "== e2e, expansion stores non-temporal=$nt"
  grep "giql_hip_inner\] H2D" gpurun_out/r04u_bench_e2e_nt$nt.err | tail -n 3
  tail -n 1 gpurun_out/r04u_bench_e2e_nt$nt.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if k.startswith('t_e2e') and 'note' not in k})"
done 2>&1 | tee gpurun_out/r04u_e2e.log
for nn in 0 1; do
  echo "== dense_40Mx400M, GIQL_HIP_NO_NARROW_BUCKETS=$nn"
  GIQL_HIP_NO_NARROW_BUCKETS=$nn timeout -k 10 400 python3 bench.py --workload dense_40Mx400M_24chrom --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r04u_dense_nn$nn.json.log 2> gpurun_out/r04u_dense_nn$nn.err || { tail -n 5 gpurun_out/r04u_dense_nn$nn.err; exit 1; }
  tail -n 1 gpurun_out/r04u_dense_nn$nn.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.3f ms, %d pairs' % (d['ms_per_step'], d['config']['pairs_per_step']), d['config']['sort'], '|', d['config']['pairs_written_by'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})"
done 2>&1 | tee gpurun_out/r04u_dense.log
timeout -k 10 500 python3 -m pytest tests/test_full_size.py -x -q -m gpu -k "dense_tables" > gpurun_out/r04u_tests4.log 2>&1
rc=$?; tail -n 3 gpurun_out/r04u_tests4.log
exit $rc
