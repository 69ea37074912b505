import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
B, T, H, dh, M = 16, 2048, 12, 64, 2048
for Kc in (2048, 4096):
    d = H * dh
    qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
    rd = torch.randn(M, d, device=dev).bfloat16()
    rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
    for _ in range(3):
        ops.relattn_fwd(qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:], rd, rwb, rrb, out, lse, **st)
    torch.cuda.synchronize()
