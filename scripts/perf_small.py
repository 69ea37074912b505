"""small HBM-bound kernels at C3 sizes: colsum, LayerNorm fwd/bwd"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
N = 32768
for d in (768, 3072, 2304):
    X = [torch.randn(N, d, device=dev).bfloat16() for _ in range(3)]
    out = torch.zeros(d, device=dev)
    i = [0]
    def f():
        i[0] += 1
        ops.colsum(X[i[0] % 3], out, N, d)
    t = timeit(f)
    print(f'colsum {N}x{d}: {t:.1f} us = {N*d*2/t/1e6:.2f} TB/s')
