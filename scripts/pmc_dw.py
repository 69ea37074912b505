"""one weight-gradient GEMM shape (ffn1 of C3, 3 K-slices) a few times, for `rocprofv3 --pmc` passes on the TT kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT, d, F = 32768, 768, 3072
X = torch.randn(NT, d, device=dev).bfloat16(); Y = torch.randn(NT, F, device=dev).bfloat16()
dW = torch.zeros(F, d, device=dev)
for i in range(4):
    ops.gemm(Y, X, dW, F, d, NT, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=3)
torch.cuda.synchronize()
