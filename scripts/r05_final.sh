#!/bin/bash
# round-5 collection: all profiles (scripts/collect_profiles.sh r05) + the three-leg bench line + the NT GEMM table
set -e -o pipefail
bash scripts/collect_profiles.sh r05
python3 bench.py --steps 10 --warmup 3 > gpurun_out/prof_r05/r05_bench_line.json 2> gpurun_out/prof_r05/bench.err || { tail -20 gpurun_out/prof_r05/bench.err; exit 1; }
python3 scripts/perf_gemm_table.py > gpurun_out/prof_r05/r05_gemm_table.txt 2>&1
ls -la gpurun_out/prof_r05
