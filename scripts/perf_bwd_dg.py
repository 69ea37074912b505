"""Attention backward at C3 with and without the dG output (how much of the query-owner kernel is its dG store traffic?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
B, T, H, dh, M, Kc = 16, 2048, 12, 64, 2048, 2048
d = H * dh
qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
rd = torch.randn(M, d, device=dev).bfloat16()
rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
q, k, v = qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:]
ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, **st)
dout = torch.randn(B, T, d, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv); delta = torch.zeros(B, H, T, device=dev)
dg = torch.empty(B, H, T, M, device=dev, dtype=torch.bfloat16)
a, c = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
for name, g, c_ in (('4-wave, with dG', dg, c), ('4-wave, no dG', None, c), ('8-wave, with dG', dg, None), ('8-wave, no dG', None, None)):
    f = lambda: ops.relattn_bwd(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                                g, a, c_, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, **st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print(f'{name}: bwd total {s.elapsed_time(e)/10:.3f} ms', flush=True)
