#!/usr/bin/env bash
# does the fill launched inside the plan (giql_hip_inner_join_dev) run slower than plan + fill?
for rep in 1 2 3; do
  for mode in fused plain; do
    if [ "$mode" = plain ]; then export GIQL_BENCH_NO_FUSE=1; else unset GIQL_BENCH_NO_FUSE; fi
    timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', 'step %.3f ms' % d['ms_per_step'], 'fill', d['roofline']['kernels']['fill']['ms'])"
  done
done
