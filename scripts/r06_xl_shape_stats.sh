#!/bin/bash
# kernel stats of a context shape: bash scripts/r06_xl_shape_stats.sh <tag> <shape>
set -e -o pipefail
ROOT=$(pwd); tag=$1; export SHAPE=$2
OUT=gpurun_out/prof_${tag}_$SHAPE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
python3 scripts/perf_xl_shape.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x -o t -- python3 scripts/perf_xl_shape.py > $OUT/stats.log 2>&1 || { tail -20 $OUT/stats.log; exit 1; }
cp $(find $OUT/x -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_${SHAPE}_kernel_stats.csv
rm -rf $OUT/x
