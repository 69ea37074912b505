#!/bin/bash
# Issue / wait anatomy of the attention kernels of one C3 layer (B = 16; forward, fused backward, dq finish, phantom dRd): two SQ PMC
# passes (the program directly after `--`, counters only), aggregated by scripts/pmc_attn.py.  Usage: bash scripts/r05_pmc_fused.sh <tag>
# -> gpurun_out/<tag>_anatomy.txt.  MXL_LIB_PATH / MXL_FUSED_NSUB pass through.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r05_pmc}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export B=16 WHICH=fused ITERS=2
run() {   # $1 = output dir, rest = counters
  local d=$1; shift
  rm -rf "$d"
  if ! rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$d" -o p -- python3 "$R/scripts/perf_attn_fused.py" > "$d.log" 2>&1; then
    echo "rocprofv3 pass failed: $d"; tail -20 "$d.log"; exit 1
  fi
}
run "$R/gpurun_out/${tag}_p1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run "$R/gpurun_out/${tag}_p2" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32
cd "$R"
python3 scripts/pmc_attn.py $(find gpurun_out/${tag}_p1 -name '*counter_collection.csv') $(find gpurun_out/${tag}_p2 -name '*counter_collection.csv') | tee gpurun_out/${tag}_anatomy.txt
rm -rf gpurun_out/${tag}_p1 gpurun_out/${tag}_p2
