#!/bin/bash
# A/B of attention-forward builds on one box: the forward's parity tests on the default library, then scripts/perf_attn_fused.py
# (per-kernel times from the library's event pairs) for every library given, two rounds in alternating order.
# Usage: bash scripts/r05_ab_fwd.sh <out-prefix> tag [tag ...]   (tag = "default" or a build/libmusicxl_<tag>.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
out=gpurun_out/$1; shift
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_relattn_fused_gpu.py -x -q -k "relattn or phantom" > ${out}_tests.log 2>&1; rc=$?
  tail -5 ${out}_tests.log
  [ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit $rc; }
fi
: > ${out}_perf.log
for round in 1 2; do
  for tag in "$@"; do
    if [ "$tag" = default ]; then lib=""; else lib="$R/symbolic_music_generation_amd/build/libmusicxl_$tag.so"; fi
    echo "== $tag (round $round)" | tee -a ${out}_perf.log
    MXL_LIB_PATH=$lib B=${B:-64} ITERS=${ITERS:-5} WHICH=fused timeout -k 10 300 python scripts/perf_attn_fused.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a ${out}_perf.log || { echo "perf run failed for $tag"; exit 1; }
  done
done
