"""One NT GEMM shape, this library and torch.matmul (hipBLASLt) a few times each: the target of the L2 counter passes
(scripts/r04_pmc_gemm.sh).  SHAPE=M,N,K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
M, N, K = (int(v) for v in os.environ.get('SHAPE', '131072,3072,768').split(','))
X = torch.randn(M, K, device=dev).bfloat16()
W = torch.randn(N, K, device=dev).bfloat16() * 0.05
Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for i in range(int(os.environ.get('ITERS', 3))):
    ops.gemm(X, W, Y, M, N, K)
    torch.matmul(X, W.t(), out=Y)
torch.cuda.synchronize()
