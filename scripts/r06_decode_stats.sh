#!/bin/bash
# kernel stats of the decode leg: bash scripts/r06_decode_stats.sh <tag> [VAR=value ...]
set -e -o pipefail
ROOT=$(pwd); tag=$1; shift
OUT=gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o t -- python3 bench.py --mode decode --no-cpu-baseline > $OUT/decode_stats.log 2>&1 || { tail -20 $OUT/decode_stats.log; exit 1; }
cp $(find $OUT/decode -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_c5_decode_kernel_stats.csv
rm -rf $OUT/decode
tail -1 $OUT/decode_stats.log | cut -c1-400
