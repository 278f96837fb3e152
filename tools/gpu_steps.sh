#!/usr/bin/env bash
# Run GPU steps in order on the gpurun box; an ordinary test failure (rc 1) lets
# later steps run, a timeout / kill / abort (rc >= 124) stops the sequence.
# usage: tools/gpu_steps.sh "name|timeout_s|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== ${name} (timeout ${tmo}s): ${cmd}"
  start=$(date +%s)
  timeout -k 10 "${tmo}" bash -c "${cmd}" > "gpurun_out/${name}.log" 2>&1
  rc=$?
  echo "=== ${name} rc=${rc} in $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/${name}.log"
  if [ "${rc}" -ge 124 ]; then echo "=== stopping after ${name} (rc ${rc})"; exit "${rc}"; fi
done
exit 0
