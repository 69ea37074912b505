for w in 5 10 20 40 1000; do echo "wg/cu $w"; MXL_DQFIN_WG_PER_CU=$w B=64 WHICH=fused ITERS=8 python scripts/perf_attn_fused.py 2>&1 | grep -o "'dqfin': [0-9.]*"; done
