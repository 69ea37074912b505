"""small HBM-bound kernels at C3 sizes: colsum, LayerNorm fwd/bwd"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
N = 32768
for d in (768, 3072, 2304):
    X = [torch.randn(N, d, device=dev).bfloat16() for _ in range(3)]
    out = torch.zeros(d, device=dev)
    i = [0]
    def f():
        i[0] += 1
        ops.colsum(X[i[0] % 3], out, N, d)
    t = timeit(f)
    print(f'colsum {N}x{d}: {t:.1f} us = {N*d*2/t/1e6:.2f} TB/s')
for d in (768, 512):
    x = [torch.randn(N, d, device=dev).bfloat16() for _ in range(3)]
    res = torch.randn(N, d, device=dev).bfloat16()
    gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    y = torch.empty(N, d, device=dev, dtype=torch.bfloat16); z = torch.empty_like(y)
    mean, rstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
    i = [0]
    def ff():
        i[0] += 1
        ops.ln_residual_fwd(x[i[0] % 3], res, gam, bet, y, z, mean, rstd, drop_p=0.1, seed=1, site=2)
    t = timeit(ff)
    print(f'ln fwd {N}x{d} (dropout): {t:.1f} us = {N*d*2*4/t/1e6:.2f} TB/s (4 streams)')
    dres = torch.empty_like(y); dx = torch.empty_like(y); dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    def fb():
        i[0] += 1
        ops.ln_residual_bwd(x[i[0] % 3], x[(i[0] + 1) % 3], z, mean, rstd, gam, dres, dx, dg, db, drop_p=0.1, seed=1, site=2)
    t = timeit(fb)
    print(f'ln bwd {N}x{d} (2 grads in, dres + dx out): {t:.1f} us = {N*d*2*5/t/1e6:.2f} TB/s (5 streams)')
V = 1190
logits = torch.randn(N, 1216, device=dev)
labels = torch.randint(4, V, (16, 2048), device=dev)
nll = torch.empty(16, 2047, device=dev); hl = torch.empty(N, 2, device=dev); acc = torch.zeros(2, device=dev)
t = timeit(lambda: ops.adaptive_nll_fwd(logits, labels, nll, hl, acc, 16, 2048, V, ()))
print(f'nll fwd {N}x{V}: {t:.1f} us = {N*V*4/t/1e6:.2f} TB/s')
