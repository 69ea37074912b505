#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel-time stats of the three bench legs, then (separate passes: the TCC counter slots
# do not hold both, and PMC must not be combined with the trace domains gpurun refuses) HBM traffic and matrix-pipe occupancy of
# the training step, and HBM traffic of an eager (non-graph) decode window.  Run on the GPU box from the repo root:
#     bash scripts/collect_profiles.sh r02
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
TRAIN="bench.py --mode train --no-cpu-baseline"
# the decode PMC window (40 eager steps) sits at the generation's mean ring occupancy (~1153 written slots), not behind the 256-token prompt
DPROMPT=${DPROMPT:-1130}
run() { local name=$1; shift; echo "== $name"; "$@" > $OUT/$name.log 2>&1 || { tail -20 $OUT/$name.log; exit 1; }; }
run train_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 $TRAIN --steps 5 --warmup 2
run decode_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o t -- python3 bench.py --mode decode --no-cpu-baseline
run reformer_stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reformer -o t -- python3 bench.py --mode reformer --no-cpu-baseline --steps 5 --warmup 2
run train_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $TRAIN --steps 2 --warmup 1
run train_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $TRAIN --steps 2 --warmup 1
run train_mfma rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o m -- python3 $TRAIN --steps 2 --warmup 1
run reformer_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rfetch -o f -- python3 bench.py --mode reformer --no-cpu-baseline --steps 2 --warmup 1
run reformer_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/rwrite -o w -- python3 bench.py --mode reformer --no-cpu-baseline --steps 2 --warmup 1
run decode_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/dfetch -o f -- python3 bench.py --mode decode --eager --decode-steps 40 --decode-prompt $DPROMPT --no-cpu-baseline
run decode_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/dwrite -o w -- python3 bench.py --mode decode --eager --decode-steps 40 --decode-prompt $DPROMPT --no-cpu-baseline
B=$(python3 -c "import bench; print(bench.WORKLOADS['c3']['B'])")
f() { find $1 -name "$2" | head -1; }
python3 scripts/pmc_traffic.py $(f $OUT/fetch '*counter_collection.csv') $(f $OUT/write '*counter_collection.csv') $OUT/${TAG}_c3_pmc_traffic.json $B
python3 scripts/pmc_mfma.py $(f $OUT/mfma '*counter_collection.csv') $(f $OUT/mfma '*kernel_trace.csv') $OUT/${TAG}_c3_mfma_util.json $B
python3 scripts/pmc_traffic.py $(f $OUT/rfetch '*counter_collection.csv') $(f $OUT/rwrite '*counter_collection.csv') $OUT/${TAG}_c4_pmc_traffic.json 16 reformer
python3 scripts/pmc_traffic.py $(f $OUT/dfetch '*counter_collection.csv') $(f $OUT/dwrite '*counter_collection.csv') $OUT/${TAG}_c5_decode_eager_pmc_traffic.json 64 decode $DPROMPT
cp $(f $OUT/train '*kernel_stats.csv') $OUT/${TAG}_c3_train_step_kernel_stats.csv
cp $(f $OUT/decode '*kernel_stats.csv') $OUT/${TAG}_c5_decode_kernel_stats.csv
cp $(f $OUT/reformer '*kernel_stats.csv') $OUT/${TAG}_c4_reformer_train_kernel_stats.csv
grep -h '^{' $OUT/train_stats.log $OUT/decode_stats.log $OUT/reformer_stats.log > $OUT/${TAG}_profiled_bench_lines.jsonl || true
# keep the merge-back small: the raw traces are large
rm -rf $OUT/train $OUT/decode $OUT/reformer $OUT/fetch $OUT/write $OUT/mfma $OUT/dfetch $OUT/dwrite $OUT/rfetch $OUT/rwrite
ls -la $OUT
