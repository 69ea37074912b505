"""one C3 attention layer, forward + backward, both backward forms on the same inputs and the same box:
   old = relattn_bwd (delta + dq8 + dkv) + add_rowbias + relattn_drd;  fused = relattn_bwd_fused (delta + fused + dq finish) + phantom dRd.
   B / KC / ITERS from the environment; prints per-kernel milliseconds from the library's own event pairs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
B, T, H, dh, M = int(os.environ.get('B', 16)), int(os.environ.get('T', 2048)), 12, 64, int(os.environ.get('M', 2048))
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
zero_mem = Kc < M + T
oph = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16) if zero_mem else None
mph = torch.empty(B, H, T, device=dev) if zero_mem else None
ITERS = int(os.environ.get('ITERS', 3))
which = os.environ.get('WHICH', 'old,fused').split(',')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = {}
for mode in which:
    if mode == 'old':
        dg = torch.empty(B, H, T, M, device=dev, dtype=torch.bfloat16)
        use_oph = zero_mem and ops.phantom_sum_applies(T=T, dh=dh, M=M, Kc=Kc)
    else:
        ws = torch.empty(ops.relattn_bwd_fused_ws_numel(B, T, H, dh, M), device=dev)
        ph = (torch.empty(int(ops.lib().mxl_relattn_drd_phantom_ws_bytes(B, T, H)), device=dev, dtype=torch.uint8)
              if (zero_mem and os.environ.get('PH_FROM_FWD', '1') == '1') else None)
    for it in range(ITERS + 1):
        if it == 1:
            ops.ktime_enable(True)
        ev[0].record()
        if mode == 'old':
            ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph if use_oph else None, mph=mph if use_oph else None, **st)
        else:
            ops.relattn_fwd(q, k, v, rd, rwb, rrb, out, lse, oph=oph, mph=mph, oph_all=True, ph_buf=ph, **st)
        ev[1].record()
        if mode == 'old':
            fin = ops.relattn_bwd(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d], dqkv[:, :, 2*d:],
                                  dg, a, c, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d, d_rd=d_rd, qr_buf=qr, defer_drd=True,
                                  oph=oph if use_oph else None, mph=mph if use_oph else None, **st)
        else:
            fin = ops.relattn_bwd_fused(q, k, v, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc-T:, :d], dqkv[:, :, d:2*d],
                                        dqkv[:, :, 2*d:], d_rd, a, c, ws, qr, dq_bs=Kc*3*d, dq_rs=3*d, dkv_bs=Kc*3*d, dkv_rs=3*d,
                                        oph=oph, mph=mph, defer_drd=True, ph_buf=ph, ph_ready=ph is not None, **st)
        ev[2].record()
        fin()
        ev[3].record()
    torch.cuda.synchronize()
    kt = ops.ktime_collect()
    ops.ktime_enable(False)
    per = {k_: round(v_[0] / max(v_[1], 1), 3) for k_, v_ in kt.items() if v_[1]}
    print(f'{mode:5s} B={B} T={T} M={M} Kc={Kc}: fwd {ev[0].elapsed_time(ev[1]):.3f} ms, bwd {ev[1].elapsed_time(ev[2]):.3f} ms, '
          f'drd {ev[2].elapsed_time(ev[3]):.3f} ms, fwd+bwd+drd {ev[0].elapsed_time(ev[3]):.3f} ms   per kernel: {per}', flush=True)
    if mode == 'old':
        del dg
    else:
        del ws
    torch.cuda.empty_cache()
