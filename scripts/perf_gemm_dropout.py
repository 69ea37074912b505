"""How much does the dropout hash in the GEMM epilogue cost?  ffn1 forward shape at C3, with / without MXL_GEMM_DROPOUT"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT, N, K = 32768, 3072, 768
X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(3)]
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
b = torch.randn(N, device=dev)
Y = [torch.empty(NT, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
def run(flags, **kw):
    for i in range(3): ops.gemm(X[i], W, Y[i], NT, N, K, flags=flags, bias=b, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30): ops.gemm(X[i % 3], W, Y[i % 3], NT, N, K, flags=flags, bias=b, **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 30 * 1e3
for _ in range(2):
    print(f'bias+relu          : {run(ops.GEMM_BIAS | ops.GEMM_RELU):.1f} us')
    print(f'bias+relu+dropout  : {run(ops.GEMM_BIAS | ops.GEMM_RELU | ops.GEMM_DROPOUT, drop_p=0.1, seed=5, site=3):.1f} us')
