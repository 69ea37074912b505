"""one C3 attention layer (B = 16): forward + backward launches, three times -- the subject of scripts/pmc_attn.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
B, T, H, dh, M, Kc = int(os.environ.get('B', 16)), 2048, 12, 64, 2048, int(os.environ.get('KC', 2048))
d = H * dh
torch.manual_seed(0)
qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
rd = torch.randn(M, d, device=dev).bfloat16()
rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
q, k, v = qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:]
dout = torch.randn(B, T, d, device=dev).bfloat16()
dqkv = torch.zeros_like(qkv); delta = torch.zeros(B, H, T, device=dev)
CH = int(os.environ.get('DG_CHUNK', B))
dg = torch.empty(CH, H, T, M, device=dev, dtype=torch.bfloat16)
a, c = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
d_rd = torch.zeros(M, d, device=dev); qr = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
OPH = bool(os.environ.get('OPH'))          # the forward's phantom value-sum for the backward (training path of the engines)
oph = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16) if OPH else None
mph = torch.empty(B, H, T, device=dev) if OPH else None
if os.environ.get('KT'):
    ops.ktime_enable(True)
for it in range(int(os.environ.get('ITERS', 3))):
    ev[0].record()
    ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph, mph=mph, **st)
    ev[1].record()
    fin = ops.relattn_bwd(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                          dg, a, c, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, d_rd=d_rd, qr_buf=qr, defer_drd=True, oph=oph, mph=mph, **st)
    ev[2].record()
    fin()
    ev[3].record()
    if CH < B:      # chunked: the dRd contraction runs inside the loop, so the two brackets are one
        pass
torch.cuda.synchronize()
kt = ops.ktime_collect() if os.environ.get('KT') else {}
print({k: round(v[0] / max(v[1], 1), 3) for k, v in kt.items()} if kt else '')
print(f'B={B} dg chunk {CH}: fwd {ev[0].elapsed_time(ev[1]):.3f} ms, bwd (delta+dq+dkv) {ev[1].elapsed_time(ev[2]):.3f} ms, drd {ev[2].elapsed_time(ev[3]):.3f} ms')
