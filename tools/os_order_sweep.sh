#!/usr/bin/env bash
# Tuning aid: onesweep tile order (GIQL_HIP_OS_ORDER) x block shape (GIQL_HIP_OS_VARIANT)
# on the bounded SEMI path (tools/scatter_probe.py, random keys).
for order in 0 2; do for variant in 0 1 3 4; do
  echo "order=$order variant=$variant"
  GIQL_HIP_OS_ORDER=$order GIQL_HIP_OS_VARIANT=$variant GIQL_PROBE_CASES=random timeout -k 10 100 python tools/scatter_probe.py 2>&1 | tail -1
done; done
