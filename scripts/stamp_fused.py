"""in-kernel stamp shares of relattn_bwd_fused_kernel (diagnostic build: bash scripts/ab_build.sh relattn_bwd_fused stamp -DMXL_STAMP;
run with MXL_LIB_PATH=symbolic_music_generation_amd/build/libmusicxl_stamp.so)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd._lib import LIB_PATH
dev = torch.device('cuda:0')
B, T, H, dh, M = int(os.environ.get('B', 16)), 2048, 12, 64, 2048
Kc = int(os.environ.get('KC', T))
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
a, c = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
d_rd = torch.zeros(M, d, device=dev); qr = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
oph = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16); mph = torch.empty(B, H, T, device=dev)
ws = torch.empty(ops.relattn_bwd_fused_ws_numel(B, T, H, dh, M), device=dev)
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * 16)()
for it in range(3):
    ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph, mph=mph, oph_all=True, **st)
    if it == 1:
        torch.cuda.synchronize(); raw.mxl_debug_fused_stamps(buf)       # reset after the warm-up
    ops.relattn_bwd_fused(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                          d_rd, a, c, ws, qr, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, oph=oph, mph=mph, **st)
torch.cuda.synchronize()
raw.mxl_debug_fused_stamps(buf)
names = ['0 tile top: stage next rows, request the rows after, park read', '1 S / dP chains issued, skew reads issued', '2 wait for chains + skew reads',
         '3 exp, dS, pack, tr fragments requested, X / Y writes, parked atomics', '4 barrier 1', '5 first units requested + dV / dK MFMAs',
         '6 dq piece: 17 pipelined units', '7 next G operands requested + slab stores', '8 dRd block MFMAs', '9 next tile G block(s)',
         '10 dRd park (one wave per tile) + new block', '11 -', '12 barrier 2', '13 final flush', '14 wait for the staged rows (vmcnt)', '15 prologue']
tot = sum(buf)
print(f'relattn_bwd_fused_kernel stamps, B={B} T={T} M={M} Kc={Kc}')
for i, n in enumerate(names):
    print(f'{n:48s} {buf[i]:16d} {100.0 * buf[i] / max(tot, 1):6.1f} %')
