#!/bin/bash
set -e -o pipefail
ROOT=$(pwd); tag=${1:-r06}
OUT=gpurun_out/dtrace_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o d -- python3 bench.py --mode decode --no-cpu-baseline --decode-steps 200 > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 scripts/trace_decode_lanes.py $f > gpurun_out/${tag}_decode_lanes.txt 2>&1 || true
rm -rf $OUT
cat gpurun_out/${tag}_decode_lanes.txt
