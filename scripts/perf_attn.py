"""attention kernel timings at the C3 layer shape (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def case(B, T, H, dh, M, Kc, bwd=True):
    d = H * dh
    qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
    rd = torch.randn(M, d, device=dev).bfloat16()
    rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
    q, k, v = qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:]
    f = lambda: ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, **st)
    ms = timeit(f)
    nbar = M if Kc == M + T else (T + 1) / 2
    fl = B * T * (2 * d * M + 4 * d * nbar)
    print(f'fwd  B={B} T={T} M={M} Kc={Kc}: {ms:.3f} ms  alg {fl/ms/1e9:.1f} TF/s', flush=True)
    if bwd:
        dout = torch.randn(B, T, d, device=dev).bfloat16()
        dqkv = torch.zeros_like(qkv); delta = torch.zeros(B, H, T, device=dev)
        dg = torch.empty(B, H, T, M, device=dev, dtype=torch.bfloat16)
        a, c = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
        g = lambda: ops.relattn_bwd(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                                    dg, a, c, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, **st)
        ms = timeit(g)
        print(f'bwd  B={B} T={T} M={M} Kc={Kc}: {ms:.3f} ms  alg {2*fl/ms/1e9:.1f} TF/s', flush=True)

case(16, 2048, 12, 64, 2048, 2048)
case(16, 2048, 12, 64, 2048, 4096, bwd=False)
