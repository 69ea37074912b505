#!/bin/bash
# Round 6: C4 Reformer leg A/B on one box over library builds (tags of symbolic_music_generation_amd/build/libmusicxl_<tag>.so, or "default"),
# alternating order, two rounds.  Usage: bash scripts/r06_rf_ab.sh <out-prefix> tag [tag ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
out=gpurun_out/$1; shift
mkdir -p gpurun_out
: > ${out}_ab.log
for round in 1 2; do
  for tag in "$@"; do
    if [ "$tag" = default ]; then lib=""; else lib="$R/symbolic_music_generation_amd/build/libmusicxl_$tag.so"; fi
    echo "== $tag (round $round)" | tee -a ${out}_ab.log
    MXL_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --mode reformer --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d = d.get('reformer', d); k = d['roofline_hbm']['kernels']; print(json.dumps({'tok_s': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 3), 'chunk_fwd_us': round(1e3 * k['chunk_fwd']['avg_launch_ms'], 1), 'chunk_bwd_q_us': round(1e3 * k['chunk_bwd_q']['avg_launch_ms'], 1), 'chunk_bwd_kv_us': round(1e3 * k['chunk_bwd_kv']['avg_launch_ms'], 1)}))" | tee -a ${out}_ab.log || { echo "bench failed for $tag"; exit 1; }
  done
done
