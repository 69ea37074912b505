#!/bin/bash
# decode leg A/B over library builds (tags of symbolic_music_generation_amd/build/libmusicxl_<tag>.so or "default"), alternating, two rounds
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
    MXL_LIB_PATH=$lib timeout -k 10 400 python3 bench.py --mode decode --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d = d.get('decode', d); print(json.dumps({'tok_s': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 4), 'full_ring_ms': round(d['full_ring']['ms_per_step'], 4), 'full_ring_frac': round(d['full_ring']['roofline']['frac'], 4)}))" | tee -a ${out}_ab.log || { echo "bench failed for $tag"; exit 1; }
  done
done
