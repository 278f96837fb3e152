#!/usr/bin/env bash
# fill time with row_a / row_b in ONE allocation vs in two, alternating processes
for rep in 1 2 3 4 5 6; do
  for mode in split joint; do
    GIQL_BENCH_OUT=$mode timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', 'step %.3f ms' % d['ms_per_step'], 'fill', d['roofline']['kernels']['fill']['ms'])"
  done
done
