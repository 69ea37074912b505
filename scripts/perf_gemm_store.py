"""Is the GEMM epilogue bound by the chip (HBM write) or by the CU?  Same per-CU work, different numbers of active CUs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
NB = 4
for K in (64, 768):
    for M, N in ((2048, 256), (8192, 256), (8192, 1024), (16384, 1024), (32768, 1024), (65536, 1024)):
        X = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
        W = torch.randn(N, K, device=dev).bfloat16() * 0.05
        Y = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
        t = timeit(lambda i: ops.gemm(X[i % NB], W, Y[i % NB], M, N, K))
        tt = timeit(lambda i: torch.matmul(X[i % NB], W.t(), out=Y[i % NB]))
        tiles = (M // 256) * (N // 256)
        print(f'K={K:4d} M={M:6d} N={N:5d} tiles={tiles:4d} out={M*N*2/1e6:6.1f} MB: ours {t*1e3:7.1f} us | torch {tt*1e3:7.1f} us', flush=True)
