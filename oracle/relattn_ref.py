"""
ORACLE (test infrastructure, NOT product code) -- dense fp32 statement of one Transformer-XL relative attention core
in *position* coordinates, the form the HIP kernels implement:

    score[i, p] = ((q_i + r_w_bias) . k_p + (q_i + r_r_bias) . Rd[i - p]) * scale,   visible iff 0 <= i - p <= M - 1

tests/test_oracle_cpu.py proves this equals the upstream formulation restated in oracle/transfoxl_ref.py
(AC/BD einsums + pad/view `_rel_shift` + same_length mask), i.e. the rel-shift identity
BD[i, j] = BDraw[i, j + qlen - 1 - i].  Both forms are pinned on an external implementation: HuggingFace XLNet's
`rel_attn_core` (tests/test_xlnet_pin_cpu.py, goldens from tests/golden/make_xlnet_relattn_goldens.py).
"""
import math

import torch


def relattn_dense(q, k, v, rd, r_w_bias, r_r_bias, M: int, scale=None, return_probs=False):
    """q (B,T,H,dh); k,v (B,Kc,H,dh) covering key positions p in [T-Kc, T) (lower positions are zero mems);
    rd (M,H,dh); biases (H,dh).  Returns out (B,T,H,dh), lse (B,H,T)."""
    B, T, H, dh = q.shape
    Kc = k.shape[1]
    scale = scale if scale is not None else 1.0 / math.sqrt(dh)
    q, k, v, rd = q.float(), k.float(), v.float(), rd.float()
    # materialise all M + T key positions p = -M .. T-1 (zeros where not stored)
    kf = torch.zeros(B, M + T, H, dh)
    vf = torch.zeros(B, M + T, H, dh)
    kf[:, M + T - Kc:] = k
    vf[:, M + T - Kc:] = v
    ac = torch.einsum('bihe,bjhe->bhij', q + r_w_bias.float(), kf)
    i = torch.arange(T)[:, None]
    p = torch.arange(-M, T)[None, :]
    dist = i - p  # (T, M+T)
    valid = (dist >= 0) & (dist <= M - 1)
    g = torch.einsum('bihe,dhe->bhid', q + r_r_bias.float(), rd)  # (B,H,T,M)
    bd = torch.gather(g, 3, dist.clamp(0, M - 1)[None, None].expand(B, H, T, M + T))
    s = (ac + bd) * scale
    s = s.masked_fill(~valid[None, None], float('-inf'))
    lse = torch.logsumexp(s, dim=-1)
    pr = torch.softmax(s, dim=-1)
    out = torch.einsum('bhij,bjhe->bihe', pr, vf)
    if return_probs:
        return out, lse, pr
    return out, lse
