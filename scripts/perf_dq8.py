import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd._lib import lib, check
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
P = lambda t: t.data_ptr() if t is not None else None
def bw(use8):
    check(lib().mxl_relattn_bwd(P(q), P(k), P(v), P(rd), P(rwb), P(rrb), P(out), P(dout), P(lse), P(delta), P(dqkv[:, Kc-T:, :d]),
          P(dqkv[:, :, d:2*d]), P(dqkv[:, :, 2*d:]), P(dg), P(a), None if use8 else P(c), B, T, H, dh, M, Kc, Kc*3*d, 3*d, Kc*3*d, 3*d, d,
          T*d, d, Kc*3*d, 3*d, Kc*3*d, 3*d, 0.125, torch.cuda.current_stream().cuda_stream), 'bwd')
for use8 in (False, True, False, True):
    for _ in range(3): bw(use8)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): bw(use8)
    e.record(); torch.cuda.synchronize()
    print('dq8' if use8 else 'dq4', f'{s.elapsed_time(e)/10:.3f} ms', flush=True)
