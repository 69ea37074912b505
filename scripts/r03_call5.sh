set -e
python3 scripts/perf_dw_tt.py > gpurun_out/dw_tt.txt 2>&1 || { tail -20 gpurun_out/dw_tt.txt; exit 1; }
MXL_GEMM_NO_TT256=1 python3 scripts/perf_dw_tt.py > gpurun_out/dw_old.txt 2>&1
cat gpurun_out/dw_tt.txt gpurun_out/dw_old.txt
python3 -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py tests/test_decode_gpu.py -q -m gpu -k "gemm or contrastive or group_beam" 2>&1 | tail -5
