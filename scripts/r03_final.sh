#!/bin/bash
# round-3 collection: all profiles + the three-leg bench line
set -e -o pipefail
bash scripts/collect_profiles.sh r03
python3 bench.py --steps 10 --warmup 3 > gpurun_out/prof_r03/r03_bench_line.json 2> gpurun_out/prof_r03/bench.err || { tail -20 gpurun_out/prof_r03/bench.err; exit 1; }
python3 scripts/perf_gemm_table.py > gpurun_out/prof_r03/gemm_table_nt.txt 2>&1
python3 scripts/perf_dw_tt.py > gpurun_out/prof_r03/gemm_table_dw_tt256.txt 2>&1
MXL_GEMM_NO_TT256=1 python3 scripts/perf_dw_tt.py > gpurun_out/prof_r03/gemm_table_dw_128.txt 2>&1
ls -la gpurun_out/prof_r03
