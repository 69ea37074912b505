#!/bin/bash
# training leg only (after a change to an attention source): kernel stats + the FETCH / WRITE / MFMA PMC passes -> gpurun_out/prof_r05t/
set -e -o pipefail
TAG=r05
OUT=gpurun_out/prof_r05t
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
TRAIN="bench.py --mode train --no-cpu-baseline"
run() { local name=$1; shift; echo "== $name"; "$@" > $OUT/$name.log 2>&1 || { tail -20 $OUT/$name.log; exit 1; }; }
run train_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 $TRAIN --steps 5 --warmup 2
run train_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $TRAIN --steps 2 --warmup 1
run train_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $TRAIN --steps 2 --warmup 1
run train_mfma rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o m -- python3 $TRAIN --steps 2 --warmup 1
B=$(python3 -c "import bench; print(bench.WORKLOADS['c3']['B'])")
f() { find $1 -name "$2" | head -1; }
python3 scripts/pmc_traffic.py $(f $OUT/fetch '*counter_collection.csv') $(f $OUT/write '*counter_collection.csv') $OUT/${TAG}_c3_pmc_traffic.json $B
python3 scripts/pmc_mfma.py $(f $OUT/mfma '*counter_collection.csv') $(f $OUT/mfma '*kernel_trace.csv') $OUT/${TAG}_c3_mfma_util.json $B
cp $(f $OUT/train '*kernel_stats.csv') $OUT/${TAG}_c3_train_step_kernel_stats.csv
grep -h '^{' $OUT/train_stats.log > $OUT/${TAG}_profiled_train_line.jsonl || true
rm -rf $OUT/train $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
