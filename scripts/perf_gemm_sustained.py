import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
M, N, K = 32768, 3072, 768
def run(nbuf, iters, tag):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nbuf)]
    W = torch.randn(N, K, device=dev).bfloat16()
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
    for i in range(5): ops.gemm(A[i % nbuf], W, C[i % nbuf], M, N, K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): ops.gemm(A[i % nbuf], W, C[i % nbuf], M, N, K)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f'{tag}: nbuf={nbuf} iters={iters}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.0f} TF/s', flush=True)
run(1, 20, 'short-hot')
run(1, 2000, 'long-hot')
run(8, 2000, 'long-cold(8 x 250MB working set)')
run(8, 40, 'short-cold')
