"""dW GEMM (dY^T X, split-K + fp32 atomics) vs number of K-slices at the C3 layer shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT, d, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 768, 3072
def timeit(fn, n=10, warm=2):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, O, K in (('qkv', 3 * d, d), ('o', d, d), ('ffn1', F, d), ('ffn2', d, F)):
    X = torch.randn(NT, K, device=dev).bfloat16(); Y = torch.randn(NT, O, device=dev).bfloat16()
    dW = torch.zeros(O, K, device=dev)
    res = []
    for ks in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 21, 24):
        t = timeit(lambda: ops.gemm(Y, X, dW, O, K, NT, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=ks))
        res.append(f'ks={ks}: {t*1e3:.0f}us')
    tiles = ((O + 127) // 128) * ((K + 127) // 128)
    print(f'{name:5s} [{O}x{K}] tiles {tiles}: ' + '  '.join(res), flush=True)
