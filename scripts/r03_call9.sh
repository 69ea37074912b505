set -e
python3 scripts/perf_gemm_table.py > gpurun_out/gt_ahead.txt 2>&1
MXL_GEMM_NOAHEAD=1 python3 scripts/perf_gemm_table.py > gpurun_out/gt_noahead.txt 2>&1
grep fwd gpurun_out/gt_ahead.txt; echo ---; grep fwd gpurun_out/gt_noahead.txt
python3 -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -3
