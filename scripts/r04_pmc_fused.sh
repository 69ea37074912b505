#!/bin/bash
# issue / wait anatomy of the fused attention backward (one C3 layer, B = 16): two PMC passes, program directly after `--`
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export B=16 WHICH=fused ITERS=2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $R/gpurun_out/pf1 -o p -- python3 $R/scripts/perf_attn_fused.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 \
  --kernel-trace --output-format csv -d $R/gpurun_out/pf2 -o p -- python3 $R/scripts/perf_attn_fused.py > /dev/null 2>&1
cd $R && python3 scripts/pmc_attn.py $(find gpurun_out/pf1 -name '*counter_collection.csv') $(find gpurun_out/pf2 -name '*counter_collection.csv')
