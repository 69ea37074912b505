"""NT GEMM epilogue variants at the bench's token count (131072), the shapes the C3 / C4 steps run: plain, bias, bias+relu+dropout,
relu-backward mask (+ fused column sums), residual add.  A/B a library build with MXL_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NB = 3


def t(fn, n=30):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


N, d, F = 131072, 768, 3072
x = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
W1 = (torch.randn(F, d, device=dev) * 0.05).bfloat16(); b1 = torch.randn(F, device=dev)
a = [torch.empty(N, F, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
fl = 2 * N * d * F
rows = []
rows.append(('ffn1 fwd plain          ', t(lambda i: ops.gemm(x[i % NB], W1, a[i % NB], N, F, d))))
rows.append(('ffn1 fwd bias+relu      ', t(lambda i: ops.gemm(x[i % NB], W1, a[i % NB], N, F, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU, bias=b1))))
rows.append(('ffn1 fwd bias+relu+drop ', t(lambda i: ops.gemm(x[i % NB], W1, a[i % NB], N, F, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU | ops.GEMM_DROPOUT,
                                                             bias=b1, drop_p=0.1, seed=5, site=3))))
W2t = (torch.randn(F, d, device=dev) * 0.05).bfloat16()           # [in = F][out = d] copy of CoreNet.3.weight (d, F)
dy = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
dF = [torch.empty(N, F, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
rows.append(('ffn2 dX plain           ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d))))
rows.append(('ffn2 dX relu-bwd mask   ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d, flags=ops.GEMM_RELU_BWD, aux=a[i % NB], alpha=1.1))))
if hasattr(ops.lib(), 'mxl_gemm_bf16_colsum'):
    cs = torch.zeros(F, device=dev)
    rows.append(('ffn2 dX mask + colsum   ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d, flags=ops.GEMM_RELU_BWD, aux=a[i % NB], alpha=1.1,
                                                                 colsum=cs))))
if hasattr(ops, 'GEMM_RELU_BWD_BITS') and ops.gemm_relu_mask_bytes(N, F):
    bits = torch.zeros(ops.gemm_relu_mask_bytes(N, F), device=dev, dtype=torch.uint8)
    rows.append(('ffn1 fwd b+r+d +savebits', t(lambda i: ops.gemm(x[i % NB], W1, a[i % NB], N, F, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU | ops.GEMM_DROPOUT |
                                                                 ops.GEMM_SAVE_RELU_MASK, aux=bits, bias=b1, drop_p=0.1, seed=5, site=3))))
    rows.append(('ffn2 dX mask bits       ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d, flags=ops.GEMM_RELU_BWD_BITS, aux=bits, alpha=1.1))))
    rows.append(('ffn2 dX bits + colsum   ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d, flags=ops.GEMM_RELU_BWD_BITS, aux=bits, alpha=1.1,
                                                                 colsum=cs))))
rows.append(('ffn2 dX residual add    ', t(lambda i: ops.gemm(dy[i % NB], W2t, dF[i % NB], N, F, d, flags=ops.GEMM_ADD_AUX, aux=a[i % NB]))))
W2 = (torch.randn(d, F, device=dev) * 0.05).bfloat16(); b2 = torch.randn(d, device=dev)
y = [torch.empty(N, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
for nm, us in rows:
    print(f'{nm} [{N}x{F}x{d}] {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
us = t(lambda i: ops.gemm(a[i % NB], W2, y[i % NB], N, d, F)); print(f'ffn2 fwd plain           [{N}x{d}x{F}] {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
us = t(lambda i: ops.gemm(a[i % NB], W2, y[i % NB], N, d, F, flags=ops.GEMM_BIAS, bias=b2)); print(f'ffn2 fwd bias            [{N}x{d}x{F}] {us:7.1f} us {fl/us/1e6:6.0f} TF/s')
