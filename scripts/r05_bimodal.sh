#!/bin/bash
# Is the ~3 % process-to-process spread of the attention kernels on one box tied to something a process can choose?
# N separate processes of scripts/perf_attn_fused.py per variant, fused / forward kernel times from the library's event pairs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() { B=64 ITERS=5 WHICH=fused timeout -k 10 200 python scripts/perf_attn_fused.py 2>&1 | grep -o "'fwd': [0-9.]*\|'fused': [0-9.]*\|'drd': [0-9.]*" | tr '\n' ' '; echo; }
for rep in 1 2 3 4 5 6; do echo -n "plain            $rep: "; run; done
for rep in 1 2 3 4; do echo -n "expandable_seg   $rep: "; PYTORCH_HIP_ALLOC_CONF=expandable_segments:True run; done
for rep in 1 2 3 4; do echo -n "no caching alloc $rep: "; PYTORCH_NO_HIP_MEMORY_CACHING=1 run; done
for rep in 1 2 3 4; do echo -n "plain again      $rep: "; run; done
