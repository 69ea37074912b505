#!/bin/bash
# round-3 baseline: the new roofline object + kernel stats of the train leg
set -e -o pipefail
OUT=gpurun_out/r03a
mkdir -p $OUT
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python3 bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_train.json 2> $OUT/bench_train.err || { tail -30 $OUT/bench_train.err; exit 1; }
cat $OUT/bench_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 bench.py --mode train --no-cpu-baseline --steps 5 --warmup 2 > $OUT/train_stats.log 2>&1 || { tail -20 $OUT/train_stats.log; exit 1; }
cp $(find $OUT/train -name '*kernel_stats.csv' | head -1) $OUT/r03a_c3_train_step_kernel_stats.csv
rm -rf $OUT/train
python3 scripts/perf_gemm_table.py > $OUT/gemm_table.txt 2>&1 || tail -5 $OUT/gemm_table.txt
tail -14 $OUT/gemm_table.txt
