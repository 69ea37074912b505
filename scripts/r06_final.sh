#!/bin/bash
# round-6 collection: all profiles (scripts/collect_profiles.sh r06), the reference's logged Reformer-base shape, the bench line
set -e -o pipefail
bash scripts/collect_profiles.sh r06
bash scripts/r06_rfbase_stats.sh r06 && cp gpurun_out/r06_rfbase_kernel_stats.csv gpurun_out/prof_r06/r06_rfbase_kernel_stats.csv
# bench.py takes `roofline.traffic` from profiles/ (and refuses a file older than the kernel sources): hand it this collection's
cp gpurun_out/prof_r06/r06_c3_pmc_traffic.json gpurun_out/prof_r06/r06_c4_pmc_traffic.json gpurun_out/prof_r06/r06_c5_decode_eager_pmc_traffic.json profiles/
python3 bench.py --steps 10 --warmup 3 > gpurun_out/prof_r06/r06_bench_line.json 2> gpurun_out/prof_r06/bench.err || { tail -20 gpurun_out/prof_r06/bench.err; exit 1; }
ls -la gpurun_out/prof_r06
