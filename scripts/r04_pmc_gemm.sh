#!/bin/bash
# L2 counters of the NT GEMM against hipBLASLt on the same operands (one pass per counter group)
set -e
OUT=gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
for SH in ${SHAPES:-131072,3072,768 131072,768,3072}; do
  export SHAPE=$SH
  echo "=== $SH"
  i=0
  for C in "${@}"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$SH/p$i -o p -- python3 scripts/perf_gemm_pmc.py > $OUT/log_$i.txt 2>&1 || { tail -5 $OUT/log_$i.txt; continue; }
    python3 scripts/pmc_sum.py $(find $OUT/$SH/p$i -name '*counter_collection.csv') gemm_nt256 Cijk
  done
done
