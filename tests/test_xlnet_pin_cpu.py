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


LAYERS = torch.load(os.path.join(ROOT, 'tests', 'golden', 'xlnet_layer.pt'))


@pytest.mark.parametrize('c', LAYERS, ids=lambda c: f"layer_q{c['qlen']}m{c['mlen']}h{c['H']}")
def test_oracle_decoder_layer_matches_xlnet_layer(c):
    """oracle DecoderLayer (qkv_net over cat(mems, h), r_net, relative attention, o_net, post-LN residual, relu FFN, post-LN)
    == a whole HuggingFace XLNetLayer carrying the same weights"""
    qlen, mlen, H, dh, B = c['qlen'], c['mlen'], c['H'], c['dh'], c['B']
    d = H * dh
    cfg = X.RefXLConfig.from_preset('debug', vocab_size=32, n_layer=1, d_model=d, n_head=H, d_head=dh, d_inner=4 * d, d_embed=d,
                                    mem_len=mlen, clamp_len=c['clamp_len'], dropout=0.0, dropatt=0.0, cutoffs=[])
    layer = X.DecoderLayer(cfg).eval()
    P = c['params']
    to_lin = lambda w: w.reshape(d, H * dh).t()            # XLNet (d_model, H, dh): y = einsum(x, w)  ->  nn.Linear weight
    with torch.no_grad():
        a = layer.dec_attn
        a.qkv_net.weight.copy_(torch.cat([to_lin(P['rel_attn.q']), to_lin(P['rel_attn.k']), to_lin(P['rel_attn.v'])], 0))
        a.r_net.weight.copy_(to_lin(P['rel_attn.r']))
        a.o_net.weight.copy_(P['rel_attn.o'].reshape(d, H * dh))      # attn_out = einsum('ibnd,hnd->ibh', vec, o)
        a.r_w_bias.copy_(P['rel_attn.r_w_bias']); a.r_r_bias.copy_(P['rel_attn.r_r_bias'])
        a.layer_norm.weight.copy_(P['rel_attn.layer_norm.weight']); a.layer_norm.bias.copy_(P['rel_attn.layer_norm.bias'])
        f = layer.pos_ff
        f.CoreNet[0].weight.copy_(P['ff.layer_1.weight']); f.CoreNet[0].bias.copy_(P['ff.layer_1.bias'])
        f.CoreNet[3].weight.copy_(P['ff.layer_2.weight']); f.CoreNet[3].bias.copy_(P['ff.layer_2.bias'])
        f.layer_norm.weight.copy_(P['ff.layer_norm.weight']); f.layer_norm.bias.copy_(P['ff.layer_norm.bias'])
        out = layer(c['h'], _pos_rows(c), c['mask'][:, :, None], c['mems'])
    assert torch.allclose(out, c['out'], atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize('V,cutoffs', [(422, [120, 300]), (1190, [200, 600, 1000]), (64, [])])
def test_oracle_adaptive_softmax_matches_torch_adaptive_log_softmax(V, cutoffs):
    """oracle ProjectedAdaptiveLogSoftmax (div_val = 1: head over [shortlist ; cluster logits], tails over the embedding slices,
    log p(j in cluster i) = head[cluster i] + tail_i[j]) against torch.nn.AdaptiveLogSoftmaxWithLoss -- an implementation of
    the same Grave et al. factorisation that this repo did not write.  torch's tails are two bias-free Linears: the first is set
    to the identity (div_value = 1 keeps the width) and the oracle's tail biases to zero, the only structural difference."""
    torch.manual_seed(V)
    d, N = 48, 40
    m = X.ProjectedAdaptiveLogSoftmax(V, d, d, cutoffs).eval()
    with torch.no_grad():
        m.out_layers[0].weight.normal_(0, 0.3)
        m.out_layers[0].bias.normal_(0, 0.3)
        if cutoffs:
            m.out_layers[0].bias[cutoffs[0]:] = 0.0
            m.cluster_weight.normal_(0, 0.3); m.cluster_bias.normal_(0, 0.3)
    hidden = torch.randn(1, N, d)
    full = m(hidden)                                            # (N, V) log-probabilities
    assert torch.allclose(full.exp().sum(-1), torch.ones(N), atol=1e-5)
    if not cutoffs:
        ref = torch.log_softmax(torch.nn.functional.linear(hidden[0], m.out_layers[0].weight, m.out_layers[0].bias), -1)
    else:
        t = torch.nn.AdaptiveLogSoftmaxWithLoss(d, V, cutoffs=cutoffs, div_value=1.0, head_bias=True).eval()
        with torch.no_grad():
            c1 = cutoffs[0]
            t.head.weight.copy_(torch.cat([m.out_layers[0].weight[:c1], m.cluster_weight], 0))
            t.head.bias.copy_(torch.cat([m.out_layers[0].bias[:c1], m.cluster_bias], 0))
            ends = cutoffs + [V]
            for i in range(len(cutoffs)):
                t.tail[i][0].weight.copy_(torch.eye(d))
                t.tail[i][1].weight.copy_(m.out_layers[0].weight[ends[i]:ends[i + 1]])
            ref = t.log_prob(hidden[0])
    assert torch.allclose(full, ref, atol=2e-5)
    # with labels: per-token NLL of the shifted targets, ignored positions exactly zero (transformer_xl.py:185-200 relies on it)
    labels = torch.randint(0, V, (1, N))
    labels[0, 5] = -100
    nll = m(hidden, labels, keep_order=True)
    tgt = labels[0, 1:]
    want = torch.where(tgt != -100, -ref[:-1].gather(1, tgt.clamp(min=0)[:, None])[:, 0], torch.zeros(N - 1))
    assert torch.allclose(nll, want, atol=2e-5) and nll[4].item() == 0.0
    # the reference calls it with keep_order=False (cluster-grouped order): same values, and its loss = mean over non-zeros
    grouped = m(hidden, labels)
    assert torch.allclose(grouped.sort().values, want.sort().values, atol=2e-5)
