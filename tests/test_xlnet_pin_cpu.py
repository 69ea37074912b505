"""The TransfoXL oracle's attention core pinned on an EXTERNAL implementation: HuggingFace XLNet's `rel_attn_core`,
`rel_shift_bnij`, `relative_positional_encoding` (installed transformers 5.15; XLNet inherits Transformer-XL's relative attention).
Goldens: tests/golden/xlnet_relattn_core.pt, made by tests/golden/make_xlnet_relattn_goldens.py (inputs + outputs only)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import transfoxl_ref as X  # noqa: E402
from oracle.relattn_ref import relattn_dense  # noqa: E402

CASES = torch.load(os.path.join(ROOT, 'tests', 'golden', 'xlnet_relattn_core.pt'))


def _pos_rows(c):
    """oracle sinusoid rows for relative positions klen-1 .. 0 (upstream pos_seq), clamped like upstream (max only)"""
    klen = c['qlen'] + c['mlen']
    pos_seq = torch.arange(klen - 1, -1, -1.0)
    if c['clamp_len'] > 0:
        pos_seq = pos_seq.clamp(max=c['clamp_len'])
    return X.PositionalEmbedding(c['H'] * c['dh'])(pos_seq)              # (klen, 1, d_model)


@pytest.mark.parametrize('c', CASES, ids=lambda c: f"q{c['qlen']}m{c['mlen']}h{c['H']}d{c['dh']}")
def test_sinusoid_and_clamp_match_xlnet(c):
    got = _pos_rows(c)[:, 0]
    ref = c['pos_emb'][1:]                   # XLNet's 'uni' sequence is klen .. 0; upstream TransfoXL's is klen-1 .. 0
    assert torch.allclose(got, ref, atol=1e-6)
    if c['clamp_len'] > 0:                   # rows beyond the clamp are identical
        far = c['qlen'] + c['mlen'] - 1 - c['clamp_len']
        assert far > 0 and torch.equal(got[0], got[far - 1])


@pytest.mark.parametrize('c', CASES, ids=lambda c: f"q{c['qlen']}m{c['mlen']}h{c['H']}d{c['dh']}")
def test_oracle_attention_module_matches_xlnet_core(c):
    """oracle RelPartialLearnableMultiHeadAttn.forward (AC, BD, pad/view rel-shift, mask, softmax, PV) with identity
    projections around it == XLNet rel_attn_core on the same heads"""
    qlen, mlen, H, dh, B = c['qlen'], c['mlen'], c['H'], c['dh'], c['B']
    klen, hd = qlen + mlen, H * dh
    m = X.RelPartialLearnableMultiHeadAttn(H, 3 * hd, dh, 0.0, 0.0, 1e-5).eval()
    rec = {}

    class Rec(torch.nn.Module):
        def forward(self, x):
            rec['attn_vec'] = x
            return torch.zeros(x.shape[0], x.shape[1], 3 * hd)
    with torch.no_grad():
        m.qkv_net.weight.copy_(torch.eye(3 * hd))
        m.r_net = torch.nn.Linear(hd, hd, bias=False)
        m.r_net.weight.copy_(c['r_weight'].reshape(hd, hd).t())
        m.r_w_bias.copy_(c['r_w_bias']); m.r_r_bias.copy_(c['r_r_bias'])
        m.o_net = Rec()
        x = torch.zeros(klen, B, 3 * hd)
        x[mlen:, :, :hd] = c['q'].reshape(qlen, B, hd)
        x[:, :, hd:2 * hd] = c['k'].reshape(klen, B, hd)
        x[:, :, 2 * hd:] = c['v'].reshape(klen, B, hd)
        m(x[mlen:], _pos_rows(c), c['mask'][:, :, None], x[:mlen])
    assert torch.allclose(rec['attn_vec'].view(qlen, B, H, dh), c['attn_vec'], atol=2e-5, rtol=1e-4)
    if 'bd_shifted' in c:                     # the pad/view rel-shift itself, on the visible band
        rr = c['q'] + c['r_r_bias']
        r_head_k = m.r_net(_pos_rows(c)).view(klen, H, dh)
        bd = X.RelPartialLearnableMultiHeadAttn._rel_shift(torch.einsum('ibnd,jnd->ijbn', rr, r_head_k))
        vis = (c['mask'] == 0)
        ref = c['bd_shifted'].permute(2, 3, 0, 1)                         # bnij -> ijbn
        assert torch.allclose(bd[vis], ref[vis], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize('c', CASES, ids=lambda c: f"q{c['qlen']}m{c['mlen']}h{c['H']}d{c['dh']}")
def test_position_coordinate_form_matches_xlnet_core(c):
    """oracle/relattn_ref.relattn_dense -- the formulation the HIP kernels implement (BD[i,p] = G[i, i-p], window of exactly M
    keys) -- against the same XLNet outputs"""
    qlen, mlen, H, dh, B = c['qlen'], c['mlen'], c['H'], c['dh'], c['B']
    klen = qlen + mlen
    r_head = torch.einsum('ih,hnd->ind', _pos_rows(c)[:, 0], c['r_weight'])      # rows = positions klen-1 .. 0
    rd = r_head.flip(0)[:mlen]                                                   # rd[d] = R(distance d), d = 0 .. M-1
    out, lse, pr = relattn_dense(c['q'].permute(1, 0, 2, 3), c['k'].permute(1, 0, 2, 3), c['v'].permute(1, 0, 2, 3), rd,
                                 c['r_w_bias'], c['r_r_bias'], M=mlen, return_probs=True)
    assert torch.allclose(out.permute(1, 0, 2, 3), c['attn_vec'], atol=2e-5, rtol=1e-4)
    if 'attn_prob' in c:
        assert torch.allclose(pr.permute(2, 3, 0, 1), c['attn_prob'], atol=1e-6)
