"""ReLU-backward-mask epilogue (ffn2 dX at C3: dF = (dD W2) * [a > 0] * scale) and aux-add epilogue"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT, N, K = 32768, 3072, 768
X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(3)]
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
A = torch.relu(torch.randn(NT, N, device=dev)).bfloat16()
Y = [torch.empty(NT, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
def run(flags, **kw):
    for i in range(3): ops.gemm(X[i], W, Y[i], NT, N, K, flags=flags, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30): ops.gemm(X[i % 3], W, Y[i % 3], NT, N, K, flags=flags, **kw)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 30 * 1e3
for _ in range(2):
    print(f'plain     : {run(0):.1f} us')
    print(f'relu_bwd  : {run(ops.GEMM_RELU_BWD, aux=A, alpha=1.1):.1f} us')
    print(f'add_aux   : {run(ops.GEMM_ADD_AUX, aux=A):.1f} us')
ref = (X[0].float() @ W.float().t()) * 1.1 * (A.float() > 0)
ops.gemm(X[0], W, Y[0], NT, N, K, flags=ops.GEMM_RELU_BWD, aux=A, alpha=1.1)
print('relu_bwd rel err', ((Y[0].float() - ref).norm() / ref.norm()).item())
ref = X[0].float() @ W.float().t() + A.float()
ops.gemm(X[0], W, Y[0], NT, N, K, flags=ops.GEMM_ADD_AUX, aux=A)
print('add_aux rel err', ((Y[0].float() - ref).norm() / ref.norm()).item())
