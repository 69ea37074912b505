"""GPU parity of the Reformer kernels against oracle/reformer_ref.py (itself pinned on HF Reformer fixtures):
bucket ids and sort permutation bit-exact (integer work), attention within bf16 tolerance, gradients vs fp32 autograd."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize('factors,n_h', [([8], 1), ([16], 2), ([4, 8], 1), ([16, 16], 2)])
def test_lsh_hash_and_sort_exact(dev, factors, n_h):
    from symbolic_music_generation_amd import ops
    from oracle.reformer_ref import lsh_buckets
    torch.manual_seed(sum(factors) + n_h)
    B, T, H, dh = 2, 256, 3, 64
    d = H * dh
    qk = bf(torch.randn(B, T, d))
    rot = torch.randn(H, dh, n_h, sum(factors) // 2)
    nb = factors[0] if len(factors) == 1 else factors
    want = lsh_buckets(qk.float().view(B, T, H, dh).transpose(1, 2), rot, nb).to(torch.int32)
    got = torch.empty(B, H, n_h * T, device=dev, dtype=torch.int32)
    ops.lsh_hash(qk.to(dev), T * d, d, rot.to(dev), got, B, T, H, dh, n_h, factors)
    agree = (got.cpu() == want).float().mean().item()
    assert agree > 0.999, agree          # fp32 summation order may flip an exact near-tie; otherwise identical
    S = n_h * T
    NB = math.prod(factors)
    bk = want.to(dev)
    sidx = torch.empty(B * H, S, device=dev, dtype=torch.int32)
    spos = torch.empty_like(sidx)
    ops.lsh_sort(bk, sidx, spos, B * H, S, T, NB * n_h)
    scaled = S * want.long() + torch.arange(S).view(1, 1, -1)
    ref = torch.argsort(scaled, dim=-1).view(B * H, S).to(torch.int32)
    assert torch.equal(sidx.cpu(), ref)                     # stable sort: bit-exact permutation
    assert torch.equal(spos.cpu(), ref % T)


def _chunk_case(dev, B, T, H, dh, n_h, lsh, drop_p=0.0, seed=0):
    from symbolic_music_generation_amd import ops
    from oracle.reformer_ref import chunked_attention
    torch.manual_seed(seed + T + dh)
    d = H * dh
    q = bf(torch.randn(B, T, d))
    k = q if lsh else bf(torch.randn(B, T, d))
    v = bf(torch.randn(B, T, d))
    S = n_h * T
    if lsh:
        buckets = torch.randint(0, 8, (B, H, n_h, T)) + 8 * torch.arange(n_h).view(1, 1, -1, 1)
        scaled = S * buckets.view(B, H, S).long() + torch.arange(S).view(1, 1, -1)
        sidx = torch.argsort(scaled, -1)
        spos = (sidx % T)
    else:
        spos = torch.arange(T).view(1, 1, T).expand(B, H, T)
    # reference in slot order
    qh = q.float().view(B, T, H, dh).transpose(1, 2).clone().requires_grad_(True)
    kh = qh if lsh else k.float().view(B, T, H, dh).transpose(1, 2).clone().requires_grad_(True)
    vh = v.float().view(B, T, H, dh).transpose(1, 2).clone().requires_grad_(True)
    g = spos.unsqueeze(-1).expand(-1, -1, -1, dh)
    if lsh:
        key = kh * torch.rsqrt(torch.mean(kh ** 2, -1, keepdim=True) + 1e-6) / math.sqrt(dh)
    else:
        key = kh / math.sqrt(dh)
    out_s, lse_s = chunked_attention(qh.gather(2, g), key.gather(2, g), vh.gather(2, g), spos, 64, self_mask=bool(lsh))
    return dict(q=q, k=k, v=v, spos=spos, qh=qh, kh=kh, vh=vh, out_s=out_s, lse_s=lse_s, d=d, S=S)


@pytest.mark.parametrize('B,T,H,dh,n_h,lsh', [(2, 256, 2, 64, 1, 0), (1, 192, 2, 64, 1, 0), (2, 256, 2, 32, 1, 0),
                                              (2, 256, 2, 64, 1, 1), (1, 256, 2, 64, 2, 1), (2, 128, 3, 32, 3, 1), (1, 128, 4, 16, 1, 1)])
def test_chunk_attention_fwd_bwd(dev, B, T, H, dh, n_h, lsh):
    from symbolic_music_generation_amd import ops
    c = _chunk_case(dev, B, T, H, dh, n_h, lsh)
    d, S, spos = c['d'], c['S'], c['spos']
    qd, kd, vd = c['q'].to(dev), c['k'].to(dev), c['v'].to(dev)
    sp = spos.to(torch.int32).contiguous().to(dev) if lsh else None
    out = torch.zeros(B, n_h, T, d, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, n_h, H, T, device=dev)
    ops.chunk_attn_fwd(qd, kd, vd, sp, out, lse, B, T, H, dh, n_h, lsh, T * d, d)
    torch.cuda.synchronize()
    # bring the reference (slot order) to (b, round, pos) order
    rnd = (torch.arange(S) // T).view(1, 1, S).expand(B, H, S)
    ref_out = torch.zeros(B, n_h, T, H, dh)
    ref_lse = torch.zeros(B, n_h, H, T)
    bi = torch.arange(B).view(B, 1, 1).expand(B, H, S)
    hi = torch.arange(H).view(1, H, 1).expand(B, H, S)
    ref_out[bi, rnd, spos, hi] = c['out_s'].detach()
    ref_lse[bi, rnd, hi, spos] = c['lse_s'].detach()
    assert (out.float().cpu().view(B, n_h, T, H, dh) - ref_out).abs().max().item() < 3e-2
    assert (lse.cpu() - ref_lse).abs().max().item() < 3e-2
    # backward: random upstream gradients on out (and on lse for multi-round LSH)
    dout = bf(torch.randn(B, n_h, T, d))
    dlse = torch.randn(B, n_h, H, T) * (1.0 if n_h > 1 else 0.0)
    do_s = dout.float().view(B, n_h, T, H, dh)[bi, rnd, spos, hi]
    dl_s = dlse[bi, rnd, hi, spos]
    (c['out_s'] * do_s).sum().add((c['lse_s'] * dl_s).sum()).backward()
    # one (T, d) slab per hash round, every element written exactly once: no zero-fill (NaN-filled here to prove it)
    dq = torch.full((B, n_h, T, d), float('nan'), device=dev); dk = dq.clone(); dv = dq.clone()
    ops.chunk_attn_bwd(qd, kd, vd, sp, out, lse, dout.to(dev), dlse.to(dev) if n_h > 1 else None, dq, dk, dv, B, T, H, dh,
                       n_h, lsh, T * d, d)
    assert not torch.isnan(dq).any() and not torch.isnan(dk).any() and not torch.isnan(dv).any()
    ref_dv = c['vh'].grad.transpose(1, 2).reshape(B, T, d)
    assert rel_err(dv.sum(1).cpu(), ref_dv) < 2e-2
    if lsh and n_h > 1:
        # the rounds summed by the key-normalisation backward (dq, dk') and on the way out (dv, bf16, a column block of a wider matrix)
        wide = torch.full((B * T, 2 * d + 8), float('nan'), device=dev, dtype=torch.bfloat16)
        ops.lsh_keynorm_bwd_rounds(qd, T * d, d, dq, dk, dv, wide, wide[:, d:], B, T, H, dh, n_h, ld_dqk=2 * d + 8, ld_dv=2 * d + 8)
        ref = c['qh'].grad.transpose(1, 2).reshape(B * T, d)
        assert rel_err(wide[:, :d].cpu(), ref) < 2e-2
        assert torch.equal(wide[:, d:2 * d], dv.sum(1).view(B * T, d).to(torch.bfloat16)) and torch.isnan(wide[:, 2 * d:].float()).all()
        one = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
        ops.lsh_keynorm_bwd(qd, T * d, d, dq.sum(1), dk.sum(1), one, B, T, H, dh)        # == the one-slab form on the summed slabs
        assert torch.equal(one, wide[:, :d].contiguous())
    elif lsh:
        dqk = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
        ops.lsh_keynorm_bwd(qd, T * d, d, dq, dk, dqk, B, T, H, dh)
        ref = c['qh'].grad.transpose(1, 2).reshape(B, T, d)
        assert rel_err(dqk.cpu(), ref) < 2e-2
    else:
        assert rel_err(dq.view(B, T, d).cpu(), c['qh'].grad.transpose(1, 2).reshape(B, T, d)) < 2e-2
        assert rel_err(dk.view(B, T, d).cpu(), c['kh'].grad.transpose(1, 2).reshape(B, T, d)) < 2e-2
    if n_h == 1:
        # one round: the bf16 destinations (strided rows of a wider matrix, as the engine passes them) hold exactly the rounded
        # f32 results, and need no zero-fill
        wide = torch.full((B * T, 3 * d + 8), float('nan'), device=dev, dtype=torch.bfloat16)
        ops.chunk_attn_bwd(qd, kd, vd, sp, out, lse, dout.to(dev), None, None, None, None, B, T, H, dh, 1, lsh, T * d, d,
                           dq16=wide, dk16=wide[:, d:], dv16=wide[:, 2 * d:], ld16=3 * d + 8)
        assert torch.equal(wide[:, :d], dq.view(B * T, d).to(torch.bfloat16))
        assert torch.equal(wide[:, d:2 * d], dk.view(B * T, d).to(torch.bfloat16))
        assert torch.equal(wide[:, 2 * d:3 * d], dv.view(B * T, d).to(torch.bfloat16))
        assert torch.isnan(wide[:, 3 * d:].float()).all()
        if lsh:
            ops.lsh_keynorm_bwd(qd, T * d, d, dq, dk, wide, B, T, H, dh, ld_dqk=3 * d + 8)
            assert torch.equal(wide[:, :d], dqk.view(B * T, d)) and torch.isnan(wide[:, 3 * d:].float()).all()


def test_chunk_attention_dropout_consistency(dev):
    """forward/backward regenerate the same keep-mask: with V = identity-like probes the dropped fraction is ~p and
    d/dV of sum(out) equals the column sums of the dropped probabilities."""
    from symbolic_music_generation_amd import ops
    B, T, H, dh = 1, 256, 1, 64
    d = H * dh
    torch.manual_seed(0)
    q, k = bf(torch.randn(B, T, d)).to(dev), bf(torch.randn(B, T, d)).to(dev)
    v = torch.ones(B, T, d, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(B, 1, T, d, device=dev, dtype=torch.bfloat16)
    out0 = torch.zeros_like(out)
    lse = torch.zeros(B, 1, H, T, device=dev)
    ops.chunk_attn_fwd(q, k, v, None, out0, lse, B, T, H, dh, 1, 0, T * d, d)
    ops.chunk_attn_fwd(q, k, v, None, out, lse, B, T, H, dh, 1, 0, T * d, d, drop_p=0.25, seed=9, site=3)
    assert (out0.float() - 1).abs().max().item() < 1e-2            # rows of P sum to 1
    rowsum = out.float()[0, 0, :, 0]                                # = sum_j keep_j P_j / 0.75
    assert 0.9 < rowsum.mean().item() < 1.1 and rowsum.std().item() > 0.01
    dq = torch.zeros(B, T, d, device=dev); dk = torch.zeros_like(dq); dv = torch.zeros_like(dq)
    ones = torch.ones(B, 1, T, d, device=dev, dtype=torch.bfloat16)
    ops.chunk_attn_bwd(q, k, v, None, out, lse, ones, None, dq, dk, dv, B, T, H, dh, 1, 0, T * d, d, drop_p=0.25, seed=9, site=3)
    # sum over keys of dV[:, e] = sum over queries of rowsum (same mask both ways)
    assert abs(dv[0, :, 0].sum().item() - rowsum.sum().item()) / rowsum.sum().item() < 1e-2


@pytest.mark.parametrize('drop_p,two', [(0.0, False), (0.1, True)])
def test_axial_embed_bwd_lds_form_at_c4_shape(dev, monkeypatch, drop_p, two):
    """mxl_axial_embed_bwd at the C4 shape (V 1190, d 512 = 128 + 384, 64 x 128 positions; HF515:222-256's embeddings backward): the
    form that accumulates 32-column slabs of the tables in LDS (round 6) against the one-atomic-per-element form -- same masks,
    sums equal up to fp32 summation order -- and, without dropout, against index_add on the host"""
    import os
    from symbolic_music_generation_amd import ops
    torch.manual_seed(5)
    B, T, V, d, A0, A1, d0 = 2, 8192, 1190, 512, 64, 128, 128
    ids = torch.randint(0, V, (B, T))
    ids[0, :7] = torch.tensor([0, V - 1, 5, 5, 5, 5, 5])
    dout = bf(torch.randn(B, T, d)); dout2 = bf(torch.randn(B, T, d)) if two else None

    def run(global_form):
        if global_form:
            monkeypatch.setenv('MXL_AXIAL_BWD_GLOBAL', '1')
        else:
            monkeypatch.delenv('MXL_AXIAL_BWD_GLOBAL', raising=False)
        dE = torch.zeros(V, d, device=dev); dW0 = torch.zeros(A0, d0, device=dev); dW1 = torch.zeros(A1, d - d0, device=dev)
        ops.axial_embed_bwd(ids.to(dev), dout.to(dev), dE, dW0, dW1, A0, A1, drop_p=drop_p, seed=11, site_emb=3, site_pos=4,
                            dout2=None if dout2 is None else dout2.to(dev))
        torch.cuda.synchronize()
        return dE.cpu(), dW0.cpu(), dW1.cpu()

    new, old = run(False), run(True)
    for a, b, name in zip(new, old, ('dE', 'dW0', 'dW1')):
        assert rel_err(a, b) < 2e-6, name
    if drop_p == 0.0:
        g = dout.float().view(-1, d)
        t = torch.arange(T)
        rE = torch.zeros(V, d).index_add_(0, ids.flatten(), g)
        rW0 = torch.zeros(A0, d0).index_add_(0, (t // A1).repeat(B), g[:, :d0])
        rW1 = torch.zeros(A1, d - d0).index_add_(0, (t % A1).repeat(B), g[:, d0:])
        assert rel_err(new[0], rE) < 1e-5 and rel_err(new[1], rW0) < 1e-5 and rel_err(new[2], rW1) < 1e-5


def test_axial_embed_and_combine(dev):
    from symbolic_music_generation_amd import ops
    torch.manual_seed(0)
    B, T, V, d, A0, A1, d0 = 2, 128, 50, 64, 8, 16, 16
    E = bf(torch.randn(V, d)); W0 = torch.randn(A0, d0); W1 = torch.randn(A1, d - d0)
    ids = torch.randint(0, V, (B, T))
    out = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
    ops.axial_embed_fwd(ids.to(dev), E.to(dev), W0.to(dev), W1.to(dev), out, A0, A1)
    t = torch.arange(T)
    pos = torch.cat([W0[t // A1], W1[t % A1]], -1)
    ref = E.float()[ids] + pos
    assert (out.float().cpu() - ref).abs().max().item() < 3e-2
    dout = bf(torch.randn(B, T, d))
    dE = torch.zeros(V, d, device=dev); dW0 = torch.zeros(A0, d0, device=dev); dW1 = torch.zeros(A1, d - d0, device=dev)
    ops.axial_embed_bwd(ids.to(dev), dout.to(dev), dE, dW0, dW1, A0, A1)
    g = dout.float()
    rE = torch.zeros(V, d).index_add_(0, ids.flatten(), g.view(-1, d))
    rW0 = torch.zeros(A0, d0).index_add_(0, (t // A1).repeat(B), g.view(-1, d)[:, :d0])
    rW1 = torch.zeros(A1, d - d0).index_add_(0, (t % A1).repeat(B), g.view(-1, d)[:, d0:])
    assert rel_err(dE.cpu(), rE) < 1e-5 and rel_err(dW0.cpu(), rW0) < 1e-5 and rel_err(dW1.cpu(), rW1) < 1e-5
    # hash-round combine + backward vs autograd
    H, dh, n_h = 2, 32, 3
    d = H * dh
    out_r = bf(torch.randn(B, n_h, T, d)); lse = torch.randn(B, n_h, H, T)
    o_r = out_r.float().requires_grad_(True); l_r = lse.clone().requires_grad_(True)
    w = torch.softmax(l_r, dim=1)                                       # (B, n_h, H, T)
    refo = (o_r.view(B, n_h, T, H, dh) * w.permute(0, 1, 3, 2).unsqueeze(-1)).sum(1).reshape(B, T, d)
    got = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
    ops.lsh_combine(out_r.to(dev), lse.to(dev), got, B, T, H, dh, n_h)
    assert (got.float().cpu() - refo).abs().max().item() < 3e-2
    dout = bf(torch.randn(B, T, d))
    refo.backward(dout.float())
    dor = torch.empty(B, n_h, T, d, device=dev, dtype=torch.bfloat16); dl = torch.empty(B, n_h, H, T, device=dev)
    ops.lsh_combine_bwd(out_r.to(dev), lse.to(dev), got, dout.to(dev), dor, dl, B, T, H, dh, n_h)
    assert rel_err(dor.cpu(), o_r.grad) < 1e-2 and rel_err(dl.cpu(), l_r.grad) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize('M,N', [(4096, 512), (1000, 264)])
def test_dropout_colsum_equals_dropout_then_colsum(dev, M, N):
    """mxl_dropout_colsum_bf16: the same masked tensor as mxl_dropout_bf16, bit for bit, and its column sums"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(31)
    x = torch.randn(M, N, device=dev).bfloat16()
    y0 = torch.empty_like(x); y1 = torch.empty_like(x)
    ops.dropout(x, y0, 0.1, seed=12, site=5)
    s0 = torch.full((N,), 1.5, device=dev); s1 = torch.full((N,), 1.5, device=dev)
    ops.colsum(y0, s0, M, N)
    ops.dropout_colsum(x, y1, s1, M, N, 0.1, 12, 5)
    assert torch.equal(y0, y1)
    assert (s0 - s1).abs().max().item() <= 1e-4 * y0.float().abs().sum(0).max().item()
    keep = (y0 != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
