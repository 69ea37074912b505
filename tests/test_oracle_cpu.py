"""CPU tests of the oracle itself: structural known-answers the reference leaves."""
import torch

from oracle.transfoxl_ref import (RefXLConfig, RefTransfoXLLMHeadModel, RelPartialLearnableMultiHeadAttn,
                                  count_parameters, cutoffs_for_vocab)
from oracle.relattn_ref import relattn_dense


def test_param_count_kat():
    # notebook/train/transformer-xl.ipynb:491 logs 92.4 M for base @ V=418
    c = RefXLConfig.from_preset('base', vocab_size=418)
    assert c.cutoffs == [] and c.mem_len == 256 and c.clamp_len == 1024 and c.d_inner == 3072
    assert count_parameters(RefTransfoXLLMHeadModel(c)) == 92_435_362


def test_cutoff_policy():
    assert cutoffs_for_vocab(418) == [] and cutoffs_for_vocab(1190) == [1000]
    assert cutoffs_for_vocab(16384) == [5000] and cutoffs_for_vocab(32768) == [10000]
    assert cutoffs_for_vocab(262144) == [20000, 40000, 200000]


def _tiny(**kw):
    torch.manual_seed(0)
    kw = dict(dict(vocab_size=97, n_layer=2, mem_len=24, clamp_len=16, cutoffs=[], dropout=0.0), **kw)
    return RefTransfoXLLMHeadModel(RefXLConfig.from_preset('debug', **kw)).eval()


def test_segmentation_invariance():
    m = _tiny()
    ids = torch.randint(0, 97, (2, 40))
    full = m(ids).prediction_scores
    mems, outs = None, []
    for s in range(0, 40, 8):
        o = m(ids[:, s:s + 8], mems=mems)
        mems = o.mems
        outs.append(o.prediction_scores)
    assert torch.allclose(full, torch.cat(outs, 1), atol=2e-5)
    mems, outs = None, []
    for s in range(40):
        o = m(ids[:, s:s + 1], mems=mems)
        mems = o.mems
        outs.append(o.prediction_scores)
    assert torch.allclose(full, torch.cat(outs, 1), atol=2e-5)


def test_logprobs_normalised_and_adaptive_consistent():
    m = _tiny(vocab_size=130, cutoffs=[100])
    ids = torch.randint(0, 130, (2, 12))
    lp = m(ids).prediction_scores
    assert torch.allclose(lp.exp().sum(-1), torch.ones(2, 12), atol=1e-4)
    lab = ids.clone()
    lab[1, 6:] = -100
    o = m(ids, labels=lab)
    # per-token NLL (cluster order) must equal -logprob[label] as a multiset; loss = mean over non-zero
    want = -lp[:, :-1].gather(2, lab[:, 1:].clamp(min=0)[..., None])[..., 0][lab[:, 1:] != -100]
    got = o.losses[o.losses != 0]
    assert torch.allclose(got.sort().values, want.sort().values, atol=1e-4)
    assert torch.allclose(o.loss, want.mean(), atol=1e-5)


def test_rel_shift_identity_and_window():
    """position-coordinate dense form == upstream einsum + pad/view rel-shift + same_length mask."""
    torch.manual_seed(1)
    H, dh, d, M, T, B = 2, 8, 16, 12, 20, 2
    att = RelPartialLearnableMultiHeadAttn(H, d, dh, 0.0, 0.0, 1e-5).eval()
    for p_ in att.parameters():
        torch.nn.init.normal_(p_, 0, 0.3)
    w = torch.randn(T, B, d)
    mems = torch.randn(M, B, d)
    klen = M + T
    pos_seq = torch.arange(klen - 1, -1, -1.0).clamp(max=7)
    inv_freq = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))
    sin_inp = torch.outer(pos_seq, inv_freq)
    pos_emb = torch.cat([sin_inp.sin(), sin_inp.cos()], -1)[:, None, :]
    ones = torch.ones(T, klen, dtype=torch.uint8)
    mask = (torch.triu(ones, 1 + M) + torch.tril(ones, 0))[:, :, None]  # mlen == mem_len -> shift 0
    with torch.no_grad():
        ref = att(w, pos_emb, mask, mems)  # LN(w + o_net(attn_vec))
        heads = att.qkv_net(torch.cat([mems, w], 0))
        q, k, v = heads.chunk(3, -1)
        q = q[-T:].view(T, B, H, dh).transpose(0, 1)
        k = k.view(klen, B, H, dh).transpose(0, 1)
        v = v.view(klen, B, H, dh).transpose(0, 1)
        dist = torch.arange(0, M).float().clamp(max=7)
        sin_d = torch.outer(dist, inv_freq)
        rd = att.r_net(torch.cat([sin_d.sin(), sin_d.cos()], -1)).view(M, H, dh)
        out, _ = relattn_dense(q, k, v, rd, att.r_w_bias, att.r_r_bias, M)
        mine = att.layer_norm(w + att.o_net(out.transpose(0, 1).reshape(T, B, H * dh)))
    assert torch.allclose(ref, mine, atol=1e-5)
    # zero mems (init_mems): dropping the stored memory rows must give the same answer as explicit zeros
    with torch.no_grad():
        zk = k.clone(); zk[:, :M] = 0
        zv = v.clone(); zv[:, :M] = 0
        a, la = relattn_dense(q, zk, zv, rd, att.r_w_bias, att.r_r_bias, M)
        b_, lb = relattn_dense(q, zk[:, M:], zv[:, M:], rd, att.r_w_bias, att.r_r_bias, M)
    assert torch.allclose(a, b_, atol=1e-6) and torch.allclose(la, lb, atol=1e-6)


def test_oracle_reproduces_its_committed_selfgolden():
    """tests/golden/xl_c1_selfgolden.pt (made by tests/golden/make_xl_selfgoldens.py) freezes the restatement's outputs: an edit
    to the oracle has to show up here"""
    import os
    import torch
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    blob = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'xl_c1_selfgolden.pt'), weights_only=False)
    m = RefTransfoXLLMHeadModel(RefXLConfig.from_preset('debug', **blob['config'])).eval()
    m.load_state_dict(blob['state_dict'])
    ids, labels = blob['ids'], blob['labels']
    with torch.no_grad():
        o1 = m(ids[:, :64], labels=labels[:, :64])
        o2 = m(ids[:, 64:], mems=o1.mems, labels=labels[:, 64:])
        gen = m.greedy_generate(ids[:, :24], max_length=88)
    assert (o1.prediction_scores - blob['logp1'].float()).abs().max().item() < 2e-2      # stored in fp16
    assert (o2.prediction_scores - blob['logp2'].float()).abs().max().item() < 2e-2
    assert abs(o1.loss.item() - blob['loss1'].item()) < 1e-5 and abs(o2.loss.item() - blob['loss2'].item()) < 1e-5
    assert torch.equal(gen, blob['greedy'])


def test_oracle_beam_search_scores_are_sequence_logprobs_and_beat_greedy():
    """the beam-search restatement (HF 4.25.1 beam_search + BeamSearchScorer): the returned score of every best hypothesis is
    its generated tokens' summed log-probability under the model divided by the sequence length, it is never below the greedy
    continuation's, and num_return_sequences hypotheses come out best first"""
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel, ref_beam_search
    torch.manual_seed(0)
    c = RefXLConfig.from_preset('debug', vocab_size=60, max_length=64, mem_len=32, cutoffs=[], n_layer=2)
    m = RefTransfoXLLMHeadModel(c).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.mul_(4)
    ids = torch.randint(4, 60, (2, 5))
    L = 24
    out, sc = ref_beam_search(m, ids, L, num_beams=3, return_scores=True, num_return_sequences=2)
    assert out.shape == (4, L) and torch.equal(out[::2, :5], ids) and torch.equal(out[1::2, :5], ids)
    greedy = m.greedy_generate(ids, L)

    def seq_logp(seq):
        with torch.no_grad():
            lp = m(seq[None, :-1]).prediction_scores[0]
        return lp[torch.arange(4, seq.numel() - 1), seq[5:]].sum().item()

    for b in range(2):
        best, second = seq_logp(out[2 * b]), seq_logp(out[2 * b + 1])
        assert abs(best / L - sc[2 * b].item()) < 1e-4 and abs(second / L - sc[2 * b + 1].item()) < 1e-4
        assert best >= second and best >= seq_logp(greedy[b]) - 1e-4
