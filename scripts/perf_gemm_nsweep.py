"""NT GEMM, time per 256 x 256 tile against N at fixed K (and against the output row stride at fixed N): is a shape slow by itself?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
M = 131072
def timeit(fn, n=10, warm=2):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for K in (768,):
    X = torch.randn(M, K, device=dev).bfloat16()
    for N, ldc in ((1536, 1536), (2048, 2048), (2304, 2304), (2304, 2560), (2304, 3072), (2560, 2560), (2816, 2816), (3072, 3072), (3072, 3328)):
        W = torch.randn(N, K, device=dev).bfloat16() * 0.05
        Yb = torch.empty(M, ldc, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(X, W, Yb, M, N, K, ldc=ldc))
        tiles = (M // 256) * ((N + 255) // 256)
        tt = timeit(lambda: torch.matmul(X, W.t(), out=Yb[:, :N])) if ldc == N else float('nan')
        print(f'K {K} N {N} ldc {ldc}: {t*1e3:7.1f} us, {2.0*M*N*K/t/1e9:6.0f} TF/s, {tiles/256:.1f} rounds, {t*1e3/(tiles/256):.2f} us per round | hipBLASLt {tt*1e3:7.1f}', flush=True)
