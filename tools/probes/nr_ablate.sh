for v in "$@"; do
  GIQL_HIP_LIB=$PWD/build/$v.so bash tools/trace_step.sh cfg5_nearest_10Mx10M_24chrom > gpurun_out/tr5_$v.log 2>&1
  echo $v; grep "k_nearest\|span" gpurun_out/tr5_$v.log
done
