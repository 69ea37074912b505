#!/bin/bash
# Multi-GPU readiness that can be shown on ONE GPU: the C3 training leg with the collective path forced at
# world size 1 (MXL_DIST_FORCE=1: per-layer buckets issued from layer_done on RCCL's stream, same stream order as at N ranks), with
# 0 / 8 / 16 compute units left free of the persistent GEMM grids (MXL_RESERVE_CUS), and a kernel trace of the forced run: which
# kernels RCCL launches and what runs beside them.  Usage: bash scripts/r05_dist_overlap.sh -> gpurun_out/r05_dist_*.{json,txt}
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
B="python3 bench.py --mode train --no-cpu-baseline --steps 10 --warmup 3"
run() { echo "== $1"; shift; env "$@" $B 2> gpurun_out/r05_dist_err.log | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('value', 'ms_per_step', 'rccl_ranks', 'grad_exchange_dtype') if k in d})" || tail -5 gpurun_out/r05_dist_err.log; }
{
run "plain (no process group)" MXL_X=0
run "forced collective path, reserve 0" MXL_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511
run "forced collective path, reserve 8" MXL_DIST_FORCE=1 MXL_RESERVE_CUS=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
run "forced collective path, reserve 16" MXL_DIST_FORCE=1 MXL_RESERVE_CUS=16 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513
run "plain again" MXL_X=0
} | tee gpurun_out/r05_dist_reserve.txt
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/r05_dist_trace"
MXL_DIST_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29514 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r05_dist_trace" -o t -- python3 "$R/bench.py" --mode train --no-cpu-baseline --steps 3 --warmup 2 > "$R/gpurun_out/r05_dist_trace.log" 2>&1 || { tail -20 "$R/gpurun_out/r05_dist_trace.log"; exit 1; }
cd "$R"
{ python3 scripts/trace_rccl.py $(find gpurun_out/r05_dist_trace -name '*kernel_trace.csv'); python3 scripts/trace_gaps.py $(find gpurun_out/r05_dist_trace -name '*kernel_trace.csv'); } | tee gpurun_out/r05_dist_trace.txt
rm -rf gpurun_out/r05_dist_trace
