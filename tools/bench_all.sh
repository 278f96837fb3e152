#!/usr/bin/env bash
# One bench.py line per BASELINE config into gpurun_out/<tag>_<workload>.json.log (copied to profiles/ afterwards).
# usage: tools/bench_all.sh <tag>
TAG="${1:-rXX}"
mkdir -p gpurun_out
for wl in cfg4_10Mx100M_24chrom cfg2_sparse_1Mx1M_1chrom cfg2_dense_1Mx1M_1chrom cfg3_semi_1Mx10M_24chrom cfg3_anti_1Mx10M_24chrom cfg3_count_1Mx10M_24chrom cfg5_nearest_10Mx10M_24chrom; do
  timeout -k 10 400 python3 bench.py --workload "$wl" --steps 10 --warmup 3 > "gpurun_out/${TAG}_${wl}.json.log" 2> "gpurun_out/${TAG}_${wl}.err"
  echo "$wl rc=$?"; tail -c 700 "gpurun_out/${TAG}_${wl}.json.log"; echo
done
