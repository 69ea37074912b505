#!/bin/bash
# decode leg only (after a change to decode.hip): kernel stats + the two PMC passes of the eager window -> gpurun_out/prof_r05d/
set -e -o pipefail
TAG=r05
OUT=gpurun_out/prof_r05d
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
run() { local name=$1; shift; echo "== $name"; "$@" > $OUT/$name.log 2>&1 || { tail -20 $OUT/$name.log; exit 1; }; }
run decode_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o t -- python3 bench.py --mode decode --no-cpu-baseline
run decode_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/dfetch -o f -- python3 bench.py --mode decode --eager --decode-steps 40 --no-cpu-baseline
run decode_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/dwrite -o w -- python3 bench.py --mode decode --eager --decode-steps 40 --no-cpu-baseline
f() { find $1 -name "$2" | head -1; }
python3 scripts/pmc_traffic.py $(f $OUT/dfetch '*counter_collection.csv') $(f $OUT/dwrite '*counter_collection.csv') $OUT/${TAG}_c5_decode_eager_pmc_traffic.json 64 decode
cp $(f $OUT/decode '*kernel_stats.csv') $OUT/${TAG}_c5_decode_kernel_stats.csv
rm -rf $OUT/decode $OUT/dfetch $OUT/dwrite
ls -la $OUT
