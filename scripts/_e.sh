set -e
MXL_LIB_PATH=symbolic_music_generation_amd/build/libmusicxl_le.so timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -2
for v in le base le base; do
  echo "== $v"
  if [ $v = base ]; then unset MXL_LIB_PATH; else export MXL_LIB_PATH=symbolic_music_generation_amd/build/libmusicxl_$v.so; fi
  SKIP_DW=1 python scripts/perf_gemm_table.py 2>&1 | grep "fwd\|dX" | sed 's/|.*//' | awk '{printf "%s %s %s us | ", $1, $2, $(NF-3)} END{print ""}'
done
