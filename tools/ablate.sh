#!/usr/bin/env bash
# timing-only ablation builds of the onesweep kernel (outputs are WRONG by design;
# run through the SEMI path, whose searches are bounded on any data)
for n in 0 1 2 3; do
  if [ $n -eq 0 ]; then lib=giql_amd/libgiql_hip.so; else lib=giql_amd/libgiql_hip_ablate$n.so; fi
  echo "ablate $n: $(GIQL_HIP_LIB=$PWD/$lib GIQL_PROBE_CASES=random timeout -k 5 120 python tools/scatter_probe.py 2>/dev/null | grep '^{')"
done
