"""LayerNorm forward / backward at the C3 shape (131072 x 768): time and effective HBM rate.  A/B builds with MXL_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
N, d = 131072, 768
NB = 3
x = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
res = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
y = [torch.empty(N, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
z = [torch.empty(N, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
dres = [torch.empty(N, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
dx = [torch.empty(N, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
mean, rstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
g, b = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
dg, db, cs = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)


def t(fn, n=30):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


us = t(lambda i: ops.ln_residual_fwd(x[i % NB], res[i % NB], g, b, y[i % NB], z[i % NB], mean, rstd, drop_p=0.1, seed=3, site=1))
print(f'ln fwd  (x, res -> y, z)              {us:7.1f} us  {4 * N * d * 2 / us / 1e6:5.2f} TB/s')
us = t(lambda i: ops.ln_residual_bwd(x[i % NB], res[i % NB], z[i % NB], mean, rstd, g, dres[i % NB], dx[i % NB], dg, db, drop_p=0.1, seed=3, site=1))
print(f'ln bwd  (dy, dy2, z -> dres, dx)      {us:7.1f} us  {5 * N * d * 2 / us / 1e6:5.2f} TB/s')
us = t(lambda i: ops.ln_residual_bwd(x[i % NB], res[i % NB], z[i % NB], mean, rstd, g, dres[i % NB], dx[i % NB], dg, db, drop_p=0.1, seed=3, site=1, dxsum=cs))
print(f'ln bwd + column sums of dx            {us:7.1f} us  {5 * N * d * 2 / us / 1e6:5.2f} TB/s')
