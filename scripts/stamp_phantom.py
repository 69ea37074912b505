"""in-kernel stamp shares of relattn_drd_phantom_kernel (diagnostic build: bash scripts/ab_build.sh relattn_drd_phantom stamp -DMXL_STAMP;
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
print('occupancy (workgroups per CU):', raw.mxl_debug_phantom_occupancy())
buf = (C.c_ulonglong * 16)()
for it in range(3):
    ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph, mph=mph, oph_all=True, **st)
    if it == 1:
        torch.cuda.synchronize(); raw.mxl_debug_phantom_stamps(buf)       # reset after the warm-up
    ops.relattn_bwd_fused(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                          d_rd, a, c, ws, qr, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, oph=oph, mph=mph, **st)
torch.cuda.synchronize()
raw.mxl_debug_phantom_stamps(buf)
names = ['0 loop top: DMA issue', '1 start values / nd reads + G(g,1)', '2 transposed reads + wait', '3 vmcnt wait + barrier (odd steps)',
         '4 exponentials (g,0)', '5 next step row / start-value reads', '6 contraction (g,0)', '7 G(g+1,0)', '8 exponentials (g,1)',
         '9 contraction (g,1)', '10 register copies', '11 -', '12 -', '13 -', '14 epilogue (atomics)', '15 prologue']
tot = sum(buf)
print(f'relattn_drd_phantom_kernel stamps, B={B} T={T} M={M} Kc={Kc}')
for i, n in enumerate(names):
    print(f'{n:48s} {buf[i]:16d} {100.0 * buf[i] / max(tot, 1):6.1f} %')
