#!/bin/bash
# A/B of fused-attention-backward builds on one box: parity tests on the default library, then scripts/perf_attn_fused.py for every
# library given, two rounds in alternating order.  A tag is <lib>[:NSUB]: lib = a tag of symbolic_music_generation_amd/build/libmusicxl_<lib>.so
# or "default" (the in-tree libmusicxl.so); NSUB sets MXL_FUSED_NSUB.  Usage: bash scripts/r05_ab_fused.sh <out-prefix> tag [tag ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
out=gpurun_out/$1; shift
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  for ns in ${TEST_NSUB:-1}; do
    MXL_FUSED_NSUB=$ns timeout -k 10 900 python -m pytest tests/test_relattn_fused_gpu.py -x -q > ${out}_tests_nsub$ns.log 2>&1; rc=$?
    echo "tests NSUB=$ns:"; tail -3 ${out}_tests_nsub$ns.log
    [ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit $rc; }
  done
fi
: > ${out}_perf.log
for round in 1 2; do
  for tag in "$@"; do
    lt=${tag%%:*}; ns=1; [ "$tag" != "$lt" ] && ns=${tag#*:}
    if [ "$lt" = default ]; then lib=""; else lib="$R/symbolic_music_generation_amd/build/libmusicxl_$lt.so"; fi
    echo "== $tag (round $round)" | tee -a ${out}_perf.log
    MXL_FUSED_NSUB=$ns MXL_LIB_PATH=$lib B=${B:-64} ITERS=${ITERS:-5} WHICH=fused timeout -k 10 300 python scripts/perf_attn_fused.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a ${out}_perf.log || { echo "perf run failed for $tag"; exit 1; }
  done
done
