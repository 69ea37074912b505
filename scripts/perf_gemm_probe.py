"""NT GEMM probe: steady-state and quantisation behaviour of the large-tile kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
NB = 4
for M, N, K in ((32768, 3072, 3072), (32768, 1024, 3072), (32768, 1024, 768), (32768, 768, 768), (32768, 768, 3072), (65536, 512, 768), (32768, 2048, 8192)):
    X = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
    W = torch.randn(N, K, device=dev).bfloat16() * 0.05
    Y = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    fl = 2.0 * M * N * K
    t = timeit(lambda i: ops.gemm(X[i % NB], W, Y[i % NB], M, N, K))
    tt = timeit(lambda i: torch.matmul(X[i % NB], W.t(), out=Y[i % NB]))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f'[{M}x{N}x{K}] tiles {tiles:5d} ({tiles/256:.2f} rounds): ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch {tt*1e3:7.1f} us {fl/tt/1e9:6.0f} TF/s', flush=True)
    del X, Y
