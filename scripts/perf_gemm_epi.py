import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
N, d, F = 32768, 768, 3072
x = torch.randn(N, d, device=dev).bfloat16(); W1 = torch.randn(F, d, device=dev).bfloat16(); b1 = torch.randn(F, device=dev)
a = torch.empty(N, F, device=dev, dtype=torch.bfloat16)
fl = 2 * N * d * F
us = t(lambda: ops.gemm(x, W1, a, N, F, d)); print(f'FFN1 plain            {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
us = t(lambda: ops.gemm(x, W1, a, N, F, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU, bias=b1)); print(f'FFN1 bias+relu        {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
us = t(lambda: ops.gemm(x, W1, a, N, F, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU | ops.GEMM_DROPOUT, bias=b1, drop_p=0.1, seed=5, site=3)); print(f'FFN1 bias+relu+drop   {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
W2 = torch.randn(d, F, device=dev).bfloat16(); dy = torch.randn(N, d, device=dev).bfloat16(); dF = torch.empty(N, F, device=dev, dtype=torch.bfloat16)
us = t(lambda: ops.gemm(dy, W2, dF, N, F, d, trans_b=True)); print(f'dX(FFN2) plain        {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
us = t(lambda: ops.gemm(dy, W2, dF, N, F, d, trans_b=True, flags=ops.GEMM_RELU_BWD, aux=a, alpha=1.1)); print(f'dX(FFN2) relu_bwd     {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
g = torch.zeros(F, d, device=dev)
for ks in (2, 3, 4, 6, 8):
    us = t(lambda: ops.gemm(dF, x, g, F, d, N, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=ks)); print(f'dW1 TT ksplits={ks}      {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
g2 = torch.zeros(d, d, device=dev)
fl2 = 2 * N * d * d
for ks in (4, 8, 14, 16):
    us = t(lambda: ops.gemm(dy, x, g2, d, d, N, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=ks)); print(f'dWo TT ksplits={ks}     {us:7.1f} us {fl2/us/1e6:6.0f} TF/s')
