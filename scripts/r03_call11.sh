set -e
B=64 KT=1 PRE=1 python3 scripts/perf_attn_layer.py 2>&1 | tail -2
B=64 KT=1 PRE=0 python3 scripts/perf_attn_layer.py 2>&1 | tail -2
python3 -m pytest tests/test_ops_gpu.py tests/test_xl_model_gpu.py tests/test_trainer_gpu.py tests/test_reformer_decode_gpu.py -x -q -m gpu 2>&1 | tail -4
