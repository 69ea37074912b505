"""Round 6: where the error of the r_net.weight gradient comes from (musicnlp/models/transformer_xl.py wraps HF TransfoXL, whose
RelPartialLearnableMultiHeadAttn forms r = r_net(pos_emb); the gradient is  dW_r = dRd^T . phi  with  dRd[dist] = sum_q dS[q, dist] (q + r_r_bias)[q]).
Pure torch, fp64 truth against emulated storage roundings of one attention layer's backward:
  dS rounded to bf16 / to scaled fp16 before the dRd contraction; q + r_r_bias rounded to bf16; P rounded to bf16 (dV only: control).
python scripts/exp_rnet_fidelity.py   (GPU if there is one)"""
import math
import os
import torch

dev = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
B, H, dh, T, M = int(os.environ.get('B', 4)), 12, 64, int(os.environ.get('T', 512)), int(os.environ.get('M', 512))
d = H * dh
std = float(os.environ.get('STD', 0.02))
K = T + M
torch.manual_seed(0)
f64 = dict(device=dev, dtype=torch.float64)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def f16s(x, bound):     # scaled fp16: one power-of-two scale from an upper bound of |x|
    k = 14 - math.ceil(math.log2(bound))
    return (x * 2.0 ** k).to(torch.float16).to(torch.float64) * 2.0 ** -k


x = torch.randn(B, T, d, **f64); mem = torch.randn(B, M, d, **f64)
Wqkv = torch.randn(3 * d, d, **f64) * std; Wr = torch.randn(d, d, **f64) * std
rwb = torch.randn(H, dh, **f64) * std; rrb = torch.randn(H, dh, **f64) * std
inv = 1.0 / (10000 ** (torch.arange(0, d, 2, **f64) / d))
pos = torch.arange(K - 1, -1, -1, **f64)
phi = torch.cat([torch.sin(pos[:, None] * inv), torch.cos(pos[:, None] * inv)], -1)       # (K, d): row K-1-dist
cat = torch.cat([mem, x], 1)
qkv = bf(bf(cat) @ bf(Wqkv).T)
q = qkv[:, M:, :d].view(B, T, H, dh); k = qkv[..., d:2 * d].view(B, K, H, dh); v = qkv[..., 2 * d:].view(B, K, H, dh)
r = bf(bf(phi) @ bf(Wr).T).view(K, H, dh)
scale = dh ** -0.5
i_ = torch.arange(T, device=dev)[:, None]; j_ = torch.arange(K, device=dev)[None, :]
dist = i_ + M - j_                                           # (T, K)
valid = (dist >= 0) & (dist <= M - 1)                        # same-length window of the reference's training mode
ridx = (K - 1 - dist).clamp(0, K - 1)
dO = torch.randn(B, T, H, dh, **f64) * 1e-3


def run(round_ds, round_qr):
    qw = bf(q + rwb); qr = q + rrb
    qr_s = bf(qr)                                            # the score side always sees bf16 (forward parity)
    AC = torch.einsum('bihe,bjhe->bhij', qw, k)
    G = torch.einsum('bihe,khe->bhik', qr_s, r)              # (B, H, T, K) by r row
    BD = torch.gather(G, 3, ridx[None, None].expand(B, H, T, K))
    S = (AC + BD) * scale
    S = S.masked_fill(~valid[None, None], float('-inf'))
    P = torch.softmax(S, -1)
    dP = torch.einsum('bihe,bjhe->bhij', dO, v)
    delta = (P * dP).sum(-1, keepdim=True)
    dS = P * (dP - delta) * scale
    dS_c = round_ds(dS)
    qr_c = round_qr(qr)
    # dRd by r row: scatter dS[q, j] onto row ridx[q, j]
    dG = torch.zeros(B, H, T, K, **f64).scatter_add_(3, ridx[None, None].expand(B, H, T, K), dS_c * valid[None, None])
    dR = torch.einsum('bhik,bihe->khe', dG, qr_c).reshape(K, d)
    return dR


def rel(a, b):
    return float((a - b).norm() / b.norm())


phic = bf(phi) - bf(phi).mean(0, keepdim=True)
ident = lambda t: t
truth = run(ident, ident)
bound = float((dO.norm(dim=-1).max() * v.norm(dim=-1).max()) * scale * 2)
print(f'B {B} T {T} M {M} std {std}: |dRd| {float(truth.norm()):.3e}, column-sum / norm {float(truth.sum(0).norm() / truth.norm()):.2e}, dS bound {bound:.2e}', flush=True)
for name, rds, rqr in (('dS bf16, Qr bf16 (the kernels)', bf, bf), ('dS bf16, Qr exact', bf, ident), ('dS exact, Qr bf16', ident, bf),
                       ('dS scaled fp16, Qr bf16', lambda t: f16s(t, bound), bf), ('dS scaled fp16, Qr exact', lambda t: f16s(t, bound), ident),
                       ('dS bf16 hi + lo, Qr bf16', lambda t: bf(t) + bf(t - bf(t)), bf)):
    dR = run(rds, rqr)
    print(f'{name:34s} dRd err {rel(dR, truth):.3e}   dW_r err: phi {rel(dR.T @ bf(phi), truth.T @ bf(phi)):.3e}   centred phi {rel(dR.T @ phic, truth.T @ phic):.3e}'
          f'   bf16(dRd) then centred {rel(bf(dR).T @ phic, truth.T @ phic):.3e}', flush=True)
