"""ln_res_bwd_kernel at the C3 step's sizes (131072 x 768) and the C4 step's (131072 x 512): microseconds and TB/s.
MXL_LIB_PATH selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
for d in (768, 512):
    N = 131072
    torch.manual_seed(0)
    NB = 4
    dy = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
    z = [torch.randn(N, d, device=dev).bfloat16() for _ in range(NB)]
    mean = torch.zeros(N, device=dev); rstd = torch.ones(N, device=dev); gam = torch.ones(d, device=dev)
    dres = torch.empty(N, d, device=dev, dtype=torch.bfloat16); dx = torch.empty_like(dres)
    dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    for variant, kw in (('plain', {}), ('dropout 0.1', dict(drop_p=0.1, seed=3, site=5))):
        run = lambda i: ops.ln_residual_bwd(dy[i % NB], None, z[i % NB], mean, rstd, gam, dres, dx, dg, db, **kw)
        for i in range(3): run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(40): run(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 40 * 1e3
        print(f'd={d} {variant:12s} {us:7.1f} us  {4 * N * d * 2 / us / 1e6:5.2f} TB/s', flush=True)
