#!/usr/bin/env bash
# timing-only onesweep builds through the bounded SEMI path (tools/scatter_probe.py): per-pass time of a 100M-row sort
# usage: tools/ablate2.sh name1 name2 ...   (build/<name>.so; "main" = the product library)
for rep in 1 2; do
for n in "$@"; do
  lib="$PWD/build/$n.so"; [ "$n" = main ] && lib="$PWD/giql_amd/libgiql_hip.so"
  echo "$n: $(GIQL_HIP_LIB=$lib GIQL_PROBE_CASES=random timeout -k 5 150 python tools/scatter_probe.py 2>/dev/null | grep '^{')"
done
done
