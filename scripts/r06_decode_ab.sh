#!/bin/bash
# Round 6: decode leg A/B on one box -- the long chain (MXL_DECODE_UNFUSED=1: 8 launches per layer) against the short chain, bench.py's
# own decode leg, alternating order, two rounds.  Usage: bash scripts/r06_decode_ab.sh <out-prefix> [variant ...]
# a variant is a list of VAR=value separated by commas (or "default").
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
out=gpurun_out/$1; shift
mkdir -p gpurun_out
: > ${out}_ab.log
[ $# -eq 0 ] && set -- MXL_DECODE_UNFUSED=1 default
for round in 1 2; do
  for v in "$@"; do
    echo "== $v (round $round)" | tee -a ${out}_ab.log
    envs=""; [ "$v" != default ] && envs=$(echo "$v" | tr ',' ' ')
    env $envs timeout -k 10 400 python3 bench.py --mode decode --no-cpu-baseline 2>/dev/null | tail -1 | \
      python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d = d.get('decode', d); print(json.dumps({'tok_s': round(d['value']), 'ms_per_step': round(d['ms_per_step'], 4), 'full_ring_ms': round(d['full_ring']['ms_per_step'], 4), 'full_ring_frac': round(d['full_ring']['roofline']['frac'], 4)}))" | tee -a ${out}_ab.log || { echo "bench failed for $v"; exit 1; }
  done
done
