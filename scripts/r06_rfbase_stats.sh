#!/bin/bash
# kernel stats of the reference's logged Reformer-base shape: bash scripts/r06_rfbase_stats.sh <tag>
set -e -o pipefail
ROOT=$(pwd); tag=$1
OUT=gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python3 scripts/perf_rfbase.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rf -o t -- python3 scripts/perf_rfbase.py > $OUT/rfbase_stats.log 2>&1 || { tail -20 $OUT/rfbase_stats.log; exit 1; }
cp $(find $OUT/rf -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_rfbase_kernel_stats.csv
rm -rf $OUT/rf
tail -1 $OUT/rfbase_stats.log
