#!/usr/bin/env bash
# round 4: giql_hip_inner with the compact plan as its default -- parity of every host-buffer test, then the e2e figures
# of the default bench line (first call, settled call, the round-3 path beside them) with the library's breakdown.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_bucket_width.py tests/test_gpu_parity.py tests/test_execute_gpu.py -x -q -m gpu -k "compact or host or execute" > gpurun_out/r04w_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/r04w_tests.log
[ $rc -eq 0 ] || exit $rc
GIQL_HIP_DEBUG_E2E=1 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04w_bench_e2e.json.log 2> gpurun_out/r04w_bench_e2e.err
rc=$?
grep "giql_hip_inner\] H2D" gpurun_out/r04w_bench_e2e.err | tail -n 6
tail -n 1 gpurun_out/r04w_bench_e2e.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if k.startswith('t_e2e') and 'note' not in k}, d['ms_per_step'])"
exit $rc
