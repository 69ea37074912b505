"""in-kernel stamp shares of relattn_fwd_kernel (diagnostic build: bash scripts/ab_build.sh relattn_fwd stamp -DMXL_STAMP;
run with MXL_LIB_PATH=symbolic_music_generation_amd/build/libmusicxl_stamp.so)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd._lib import LIB_PATH
dev = torch.device('cuda:0')
B, T, H, dh, M = int(os.environ.get('B', 64)), 2048, 12, 64, 2048
Kc = int(os.environ.get('KC', T))
d = H * dh
torch.manual_seed(0)
qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
rd = torch.randn(M, d, device=dev).bfloat16()
rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
q, k, v = qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:]
zero_mem = Kc < M + T
oph = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16) if zero_mem else None
mph = torch.empty(B, H, T, device=dev) if zero_mem else None
raw = C.CDLL(LIB_PATH)
buf = (C.c_ulonglong * 16)()
for it in range(3):
    if it == 1:
        torch.cuda.synchronize(); raw.mxl_debug_fwd_stamps(buf)       # reset after the warm-up
    ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph, mph=mph, oph_all=zero_mem, **st)
torch.cuda.synchronize()
raw.mxl_debug_fwd_stamps(buf)
names = ['0 tile: next K / V / Rd rows requested', '1 tile: Rd fragments read, G chains issued', '2 tile: G -> fp16, skew writes',
         '3 tile: K fragments read, S chains issued', '4 tile: skew reads + S + BD (waits for both)', '5 tile: exp2, row sums',
         '6 tile: P -> bf16, V^T reads, PV MFMAs issued', '7 tile: barrier 1', '8 tile: staging stores', '9 tile: barrier 2',
         '10 phantom: staging store / request', '11 phantom: Rd fragments, G chains (+ mask)', '12 phantom: exp2, row sums (waits for the chains)',
         '13 phantom: P -> bf16, Rd^T reads, oph MFMAs issued', '14 phantom: barrier', '15 prologue, epilogue, loop control']
tot = sum(buf)
print(f'relattn_fwd_kernel stamps, B={B} T={T} M={M} Kc={Kc}')
for i, n in enumerate(names):
    print(f'{n:60s} {buf[i]:16d} {100.0 * buf[i] / max(tot, 1):6.1f} %')
