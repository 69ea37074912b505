"""Decode parity: greedy token-for-token vs the CPU oracle's HF-style loop (prompt, then one token at a time with mems),
hipGraph replay == eager, and the sampler's filtered distribution vs HF's temperature/top-k/top-p/renormalise recipe."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(dev, **kw):
    from tests.test_xl_model_gpu import _pair as p
    return p(dev, **kw)


@pytest.mark.parametrize('use_graph', [False, True])
def test_greedy_token_parity(dev, use_graph):
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=160, seed=11)
    ref.eval(); m.eval()
    prompt = torch.randint(4, 1190, (3, 24))
    want = ref.greedy_generate(prompt, max_length=120)          # crosses the mem_len=64 ring boundary
    got = m.generate(input_ids=prompt.to(dev), max_length=120, do_sample=False, use_graph=use_graph).cpu()
    assert got.shape == want.shape
    mism = (got != want).nonzero()
    # bf16 argmax ties can flip a token; report the first divergence.  Must match over the compared horizon.
    assert mism.numel() == 0, f'first divergence at {mism[0].tolist()}: got {got[tuple(mism[0])]} want {want[tuple(mism[0])]}'


def test_decode_logprobs_match_full_forward(dev):
    """step-wise decode (ring cache) == one-shot forward on the same tokens (segmentation invariance on the HIP path)."""
    from symbolic_music_generation_amd.generate import XLDecoder
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=128, seed=5)
    m.eval()
    ids = torch.randint(4, 1190, (2, 100), device=dev)
    full = m(input_ids=ids).prediction_scores          # (2, 100, V)
    dec = XLDecoder(m.engine, 2, 128)
    samp = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0)
    dec.prefill(ids[:, :10], samp)
    errs = []
    for t in range(10, 100):
        dec.force_tokens(ids[:, t])                    # teacher forcing: replace the sampled token
        dec.step(samp, want_logp=True)
        errs.append((dec.logp - full[:, t]).abs().max().item())
    assert max(errs) < 4e-2, max(errs)


def test_sampler_distribution(dev):
    from symbolic_music_generation_amd import ops
    torch.manual_seed(0)
    B, V = 4, 1190
    logp = torch.log_softmax(torch.randn(B, V) * 2, -1)
    lp = logp.to(dev)
    ids = torch.zeros(B, 8, device=dev, dtype=torch.int64)
    t = torch.zeros(1, device=dev, dtype=torch.int32)
    rng = torch.zeros(1, device=dev, dtype=torch.int64)
    probs = torch.zeros(B, V, device=dev)
    for (k, p_, temp) in [(8, 1.0, 1.0), (0, 0.9, 0.7), (50, 0.5, 1.3)]:
        ops.sample(lp, ids, t, rng, 123, do_sample=True, top_k=k, top_p=p_, temperature=temp, out_probs=probs)
        # HF: TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> LogitNormalization -> softmax
        s = logp / temp
        if k:
            kth = s.topk(k, -1).values[:, -1:]
            s = s.masked_fill(s < kth, float('-inf'))
        if p_ < 1.0:
            sl, si = s.sort(-1, descending=False)
            cp = sl.softmax(-1).cumsum(-1)
            rm = cp <= (1 - p_)
            rm[:, -1:] = False
            s = s.masked_fill(rm.scatter(1, si, rm), float('-inf'))
        want = s.softmax(-1)
        got = probs.cpu()
        assert (got - want).abs().max().item() < 1e-5, (k, p_, temp)
        tok = ids[:, 1].cpu()
        assert (want.gather(1, tok[:, None]) > 0).all()
    # empirical frequencies follow the distribution (top_k = 4: chi-square-ish bound)
    counts = torch.zeros(V)
    ops.sample(lp, ids, t, rng, 7, do_sample=True, top_k=4, top_p=1.0, temperature=1.0, out_probs=probs)
    want = probs[0].cpu()
    n = 2000
    for i in range(n):
        rng.fill_(i)
        ops.sample(lp, ids, t, rng, 7, do_sample=True, top_k=4, top_p=1.0, temperature=1.0)
        counts[ids[0, 1].item()] += 1
    sel = want > 0
    assert sel.sum().item() == 4
    assert ((counts[sel] / n) - want[sel]).abs().max().item() < 0.05
    # greedy = argmax
    ops.sample(lp, ids, t, rng, 7, do_sample=False)
    assert torch.equal(ids[:, 1].cpu(), logp.argmax(-1))


def test_sampler_repetition_penalty_and_typical_p_vs_hf_processors(dev):
    """The remaining keys of the reference's `sample` strategy (eval.py:279): the sampler's filtered distribution against HF's
    own RepetitionPenaltyLogitsProcessor / TypicalLogitsWarper (the installed `transformers`, an external implementation),
    chained in GenerationMixin's order: penalty -> temperature -> top-k -> top-p -> typical-p -> softmax."""
    lp_mod = pytest.importorskip('transformers.generation.logits_process')
    from symbolic_music_generation_amd import ops
    torch.manual_seed(1)
    B, V, Th = 4, 1190, 40
    logp = torch.log_softmax(torch.randn(B, V) * 2.5, -1)
    lp = logp.to(dev)
    hist = torch.randint(0, V, (B, Th))
    hist[:, 5] = hist[:, 6]                                  # a repeated id is penalised once
    hist[0, :8] = logp[0].topk(8).indices                    # make sure the penalty hits the head of the distribution
    ids = torch.zeros(B, Th + 8, dtype=torch.int64); ids[:, :Th] = hist
    ids_d = ids.to(dev)
    t = torch.full((1,), Th - 1, device=dev, dtype=torch.int32)
    rng = torch.zeros(1, device=dev, dtype=torch.int64)
    probs = torch.zeros(B, V, device=dev)
    cases = [dict(rp=1.3, typ=1.0, k=0, p=1.0, temp=1.0), dict(rp=1.0, typ=0.6, k=0, p=1.0, temp=1.0),
             dict(rp=1.2, typ=0.9, k=64, p=0.9, temp=0.8), dict(rp=1.0, typ=0.2, k=16, p=1.0, temp=1.4),
             dict(rp=1.5, typ=0.95, k=0, p=0.7, temp=1.0)]
    for c in cases:
        ops.sample(lp, ids_d, t, rng, 5, do_sample=True, top_k=c['k'], top_p=c['p'], temperature=c['temp'],
                   repetition_penalty=c['rp'], typical_p=c['typ'], out_probs=probs)
        s = logp.clone()
        if c['rp'] != 1.0:
            s = lp_mod.RepetitionPenaltyLogitsProcessor(c['rp'])(hist, s)
        if c['temp'] != 1.0:
            s = lp_mod.TemperatureLogitsWarper(c['temp'])(hist, s)
        if c['k']:
            s = lp_mod.TopKLogitsWarper(c['k'])(hist, s)
        if c['p'] < 1.0:
            s = lp_mod.TopPLogitsWarper(c['p'])(hist, s)
        if c['typ'] < 1.0:
            s = lp_mod.TypicalLogitsWarper(mass=c['typ'])(hist, s)
        want = s.softmax(-1)
        got = probs.cpu()
        assert ((got > 0) == (want > 0)).all(), c
        assert (got - want).abs().max().item() < 2e-5, c
        tok = ids_d[:, Th].cpu()
        assert (want.gather(1, tok[:, None]) > 0).all(), c
    # greedy with a penalty: argmax of the penalised scores
    ops.sample(lp, ids_d, t, rng, 5, do_sample=False, repetition_penalty=5.0)
    want = lp_mod.RepetitionPenaltyLogitsProcessor(5.0)(hist, logp.clone()).argmax(-1)
    assert torch.equal(ids_d[:, Th].cpu(), want)
    assert (want[0] != logp[0].argmax()).item()             # the penalty did move row 0's choice


def test_generate_with_repetition_penalty_runs_in_graph(dev):
    """model.generate(..., repetition_penalty, typical_p) end to end: eager == hipGraph replay (the penalty reads the id
    history through the device-side step counter, so it must stay correct under replay)"""
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=96, seed=3)
    m.eval()
    prompt = torch.randint(4, 1190, (2, 12)).to(dev)
    kw = dict(max_length=80, do_sample=True, top_k=32, top_p=0.9, typical_p=0.8, repetition_penalty=1.3)
    a = m.generate(input_ids=prompt, use_graph=False, **kw).cpu()
    m._decoder.rng.zero_()                                   # the draw counter runs on across calls; rewind it
    b = m.generate(input_ids=prompt, use_graph=True, **kw).cpu()
    assert torch.equal(a, b)
    g0 = m.generate(input_ids=prompt, max_length=80, do_sample=False).cpu()
    g1 = m.generate(input_ids=prompt, max_length=80, do_sample=False, repetition_penalty=3.0).cpu()
    assert not torch.equal(g0, g1)
    # with a strong penalty greedy decoding repeats less
    rep = lambda x: sum(len(r.tolist()) - len(set(r.tolist())) for r in x[:, 12:])
    assert rep(g1) <= rep(g0)


def test_decode_qkv_fused_equals_projection_plus_append(dev):
    """mxl_decode_qkv == mxl_gemm_skinny_bf16 + mxl_kv_append, bit for bit: qkv buffer, both rings (only the slot of step t
    touched), and q + r_r_bias"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(4)
    B, d, H, dh, M = 7, 128, 8, 16, 40
    x = (torch.randn(B, d)).bfloat16().to(dev)
    w = (torch.randn(3 * d, d) * 0.1).bfloat16().to(dev)
    rrb = torch.randn(d, device=dev) * 0.2
    for t in (0, 13, 39, 40, 97):                      # 97 % 40: the ring wraps
        t_dev = torch.full((1,), t, device=dev, dtype=torch.int32)
        kc0 = torch.randn(B, H, M, dh, device=dev).bfloat16(); vc0 = torch.randn(B, H, M, dh, device=dev).bfloat16()
        kc1, vc1, kc2, vc2 = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
        qkv1 = torch.empty(B, 3 * d, device=dev, dtype=torch.bfloat16); qkv2 = torch.empty_like(qkv1)
        qr1 = torch.empty(B, d, device=dev, dtype=torch.bfloat16); qr2 = torch.empty_like(qr1)
        ops.gemm_skinny(x, w, qkv1, B, 3 * d, d)
        ops.kv_append(qkv1, kc1, vc1, t_dev, rrb=rrb, qr_out=qr1)
        ops.decode_qkv(x, w, qkv2, kc2, vc2, t_dev, rrb, qr2, dh)
        assert torch.equal(qkv1, qkv2) and torch.equal(qr1, qr2)
        assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
        slot = t % M
        untouched = torch.ones(M, dtype=torch.bool); untouched[slot] = False
        assert torch.equal(kc2[:, :, untouched], kc0[:, :, untouched]) and not torch.equal(kc2[:, :, slot], kc0[:, :, slot])


@pytest.mark.parametrize('B,H,dh,M,t,pieces', [(3, 2, 64, 512, 40, 2), (3, 2, 64, 512, 299, 4), (2, 3, 64, 512, 5000, 2),
                                               (2, 2, 64, 512, 5000, 3), (5, 1, 64, 256, 0, 4), (2, 2, 32, 128, 77, 2),
                                               (2, 2, 64, 2048, 1152, 8)])
def test_decode_attention_ring_pieces(dev, B, H, dh, M, t, pieces):
    """mxl_relattn_decode_split: every (sequence, head) ring cut into `pieces` workgroups, merged by the last arrival in piece
    order.  Against fp64 single-query attention over the ring (written slots + zero memories) and against the one-piece kernel
    (the probabilities are rounded to bf16 relative to the piece's maximum: last-bit differences only); two launches in a row
    give identical bits (the arrival counters come back to zero, the merge order is fixed)."""
    from symbolic_music_generation_amd import ops
    from symbolic_music_generation_amd.ops import lib, _p, _stream, check
    torch.manual_seed(B * 100 + t)
    d = H * dh
    kc = torch.randn(B, H, M, dh, device=dev).bfloat16(); vc = torch.randn(B, H, M, dh, device=dev).bfloat16()
    nvalid = min(t + 1, M)
    if nvalid < M:                       # never-written slots are zero memories
        order = torch.arange(M, device=dev)
        kc[:, :, nvalid:] = 0; vc[:, :, nvalid:] = 0
    qkv = torch.randn(B, 3 * d, device=dev).bfloat16()
    bd = torch.randn(B, H, M, device=dev) * 2
    rwb = torch.randn(H, dh, device=dev) * .3
    t_dev = torch.tensor([t], device=dev, dtype=torch.int32)
    scale = dh ** -0.5
    one = torch.empty(B, d, device=dev, dtype=torch.bfloat16); out = torch.empty_like(one); out2 = torch.empty_like(one)
    check(lib().mxl_relattn_decode(_p(qkv), _p(kc), _p(vc), _p(bd), _p(rwb), _p(one), _p(t_dev), B, H, dh, M, scale, _stream()), 'decode')
    ws, arrived = ops.relattn_decode_split_scratch(B, H, dh, pieces, dev)
    for o in (out, out2):
        check(lib().mxl_relattn_decode_split(_p(qkv), _p(kc), _p(vc), _p(bd), _p(rwb), _p(o), _p(t_dev), B, H, dh, M, scale, pieces,
                                             _p(ws), _p(arrived), _stream()), 'decode split')
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and int(arrived.abs().sum().item()) == 0
    # fp64 reference: slot s holds position with distance (t mod M - s) mod M
    q = (qkv[:, :d].float().view(B, H, dh) + rwb).bfloat16().double()
    tm = t % M
    dist = (tm - torch.arange(M, device=dev)) % M
    sc = (torch.einsum('bhe,bhse->bhs', q, kc.double()) + bd.double().gather(2, dist[None, None].expand(B, H, M))) * scale
    pr = torch.softmax(sc, dim=-1)
    ref = torch.einsum('bhs,bhse->bhe', pr, vc.double()).reshape(B, d)
    assert (out.double() - ref).abs().max().item() < 2e-2
    assert (out.float() - one.float()).abs().max().item() <= 2e-2 and (one.double() - ref).abs().max().item() < 2e-2


def test_ring_pieces_in_the_decoder_graph_equals_eager(dev, monkeypatch):
    """the decoder with every ring cut into pieces (forced: the tiny model's rings are short): hipGraph replay == eager launches,
    token for token -- the arrival counters return to zero after every launch, the merge order is fixed -- and the first tokens
    are the one-piece decoder's (last-bit differences of the attention output do not move a greedy choice that is not a near-tie)"""
    from symbolic_music_generation_amd.generate import XLDecoder
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=128, seed=31)
    m.eval()
    g = torch.Generator().manual_seed(32)
    prompt = torch.randint(4, 1190, (3, 10), generator=g).to(dev)
    monkeypatch.setenv('MXL_DECODE_PIECES', '2')
    d2 = XLDecoder(m.engine, 3, 128, seed=3)
    assert d2.pieces == 2 and d2.split is not None
    a = d2.generate(prompt, 90, do_sample=False, use_graph=True)
    b = XLDecoder(m.engine, 3, 128, seed=3).generate(prompt, 90, do_sample=False, use_graph=False)
    assert torch.equal(a, b) and int(d2.split[1].abs().sum().item()) == 0
    # and the tokens are the oracle's as long as the oracle's own margin is not a near-tie
    monkeypatch.setenv('MXL_DECODE_PIECES', '1')
    one = XLDecoder(m.engine, 3, 128, seed=3).generate(prompt, 90, do_sample=False, use_graph=True)
    same = (a == one).all(dim=0).float()
    assert same[:30].all(), 'ring pieces change a greedy token within the first 20 generated positions'


@pytest.mark.parametrize('B,H,M', [(64, 12, 2048), (5, 2, 300), (17, 3, 301), (1, 1, 64)])
def test_decode_bd_vs_einsum(dev, B, H, M):
    """mxl_decode_bd == einsum('bhe,rhe->bhr') of the bf16 operands in fp32 (HF: BD = einsum("ibnd,jnd->ijbn", rr_head_q,
    r_head_k) for one query), and == the batched GEMM it replaces on the decode path; rows / distances beyond B / M untouched"""
    from symbolic_music_generation_amd import ops
    from symbolic_music_generation_amd._lib import lib
    torch.manual_seed(B + M)
    dh, d = 64, H * 64
    qr = torch.randn(B, d, device=dev).bfloat16()
    rd = torch.randn(M, d, device=dev).bfloat16()
    bd = torch.full((B + 1, H, M), 7.0, device=dev, dtype=torch.float32)
    ops.check(lib().mxl_decode_bd(qr.data_ptr(), rd.data_ptr(), bd.data_ptr(), B, H, dh, M, d, d, None), 'mxl_decode_bd')
    torch.cuda.synchronize()
    ref = torch.einsum('bhe,rhe->bhr', qr.float().view(B, H, dh), rd.float().view(M, H, dh))
    assert torch.allclose(bd[:B], ref, rtol=1e-5, atol=1e-4)
    assert torch.all(bd[B] == 7.0)
    bd2 = torch.empty(B, H, M, device=dev, dtype=torch.float32)
    ops.gemm_batched(qr, rd, bd2, B, M, dh, lda=d, ldb=d, ldc=H * M, flags=ops.GEMM_OUT_F32, batch=H, bdiv=1,
                     sA=(dh, 0), sB=(dh, 0), sC=(M, 0))
    assert torch.allclose(bd[:B], bd2, rtol=1e-6, atol=1e-5)
    assert lib().mxl_decode_bd(qr.data_ptr(), rd.data_ptr(), bd.data_ptr(), 65, H, dh, M, d, d, None) < 0     # B > 64: argument error


def test_beam_search_matches_oracle(dev):
    """`model.generate(num_beams=3)` (the reference's 'beam' strategy with do_sample=False, eval.py:302-321) against the oracle's
    restatement of HF 4.25.1 beam_search + BeamSearchScorer: same best hypothesis per prompt, token for token, and the same
    length-normalised score.  Weights 3x the init so that beams really compete (at the init scale the distribution is flat);
    a fork is accepted only where the oracle's own two best continuations are within bf16 noise of each other."""
    from oracle.transfoxl_ref import ref_beam_search
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=160, seed=13)
    ref.eval(); m.eval()
    g = torch.Generator().manual_seed(14)
    prompt = torch.randint(4, 1190, (2, 20), generator=g)
    L = 90                                                          # crosses the mem_len = 64 ring boundary
    want, w_sc = ref_beam_search(ref, prompt, L, num_beams=3, early_stopping=True, return_scores=True)
    from symbolic_music_generation_amd.generate import XLDecoder, beam_search
    dec = XLDecoder(m.engine, 2 * 3, L)
    got, g_sc = beam_search(dec, prompt.to(dev), L, num_beams=3, early_stopping=True, return_scores=True)
    got = got.cpu()
    assert got.shape == want.shape and torch.equal(got[:, :20], prompt)
    print('beam scores (sum log p / len): HIP', g_sc.tolist(), 'oracle', w_sc.tolist())
    assert (g_sc - w_sc).abs().max().item() < 2e-2
    same = (got == want).all(1)
    if not same.all():          # a different hypothesis may win only if it scores the same to within the bf16 envelope
        for b in (~same).nonzero().flatten().tolist():
            lp = ref(got[b:b + 1, :-1]).prediction_scores[0]
            s_got = lp[torch.arange(19, L - 1), got[b, 20:]].sum().item() / L
            assert abs(s_got - w_sc[b].item()) < 5e-3, (b, s_got, w_sc[b].item())
    # through the public API, two hypotheses per prompt
    out = m.generate(input_ids=prompt.to(dev), max_length=60, num_beams=3, num_return_sequences=2, early_stopping=True).cpu()
    want2 = ref_beam_search(ref, prompt, 60, num_beams=3, early_stopping=True, num_return_sequences=2)
    assert out.shape == want2.shape == (4, 60)
    assert (out == want2).float().mean().item() > 0.9


def test_beam_sample_and_return_sequences(dev):
    """beam_sample (the reference's default for strategy='beam': do_sample=True, eval.py:318) and sampling with
    num_return_sequences: shapes, prompts kept, ids inside the vocabulary, reproducible under the same seed"""
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=160, seed=13)
    m.eval()
    prompt = torch.randint(4, 1190, (2, 12), device=dev)
    kw = dict(input_ids=prompt, max_length=50, num_beams=3, do_sample=True, top_k=16, temperature=0.9, renormalize_logits=True,
              early_stopping=True)
    a = m.generate(**kw, num_return_sequences=2, seed=5)
    b = m.generate(**kw, num_return_sequences=2, seed=5)
    c = m.generate(**kw, num_return_sequences=2, seed=6)
    assert a.shape == (4, 50) and torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(a[:, :12], prompt.repeat_interleave(2, 0)) and (a >= 0).all() and (a < 1190).all()
    s = m.generate(input_ids=prompt, max_length=40, do_sample=True, num_return_sequences=3)        # HF default top_k = 50
    assert s.shape == (6, 40) and torch.equal(s[:, :12], prompt.repeat_interleave(3, 0))
    with pytest.raises(ValueError, match='num_return_sequences'):
        m.generate(input_ids=prompt, max_length=40, num_return_sequences=2)


def test_group_beam_search_matches_oracle(dev):
    """Diverse (group) beam search -- `generate(num_beams=4, num_beam_groups=2, diversity_penalty=...)`, the reference's 'beam'
    strategy with num_beam_groups (musicnlp/trainer/eval.py:303-317) -- against the oracle's restatement of HF 4.25.1
    group_beam_search + HammingDiversityLogitsProcessor: the same best hypothesis per prompt (a fork only where its score equals
    the oracle's to bf16 noise), the same length-normalised scores, and the diversity penalty really separates the groups."""
    from oracle.transfoxl_ref import ref_group_beam_search
    from symbolic_music_generation_amd.generate import XLDecoder, group_beam_search
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=160, seed=13)
    ref.eval(); m.eval()
    g = torch.Generator().manual_seed(15)
    prompt = torch.randint(4, 1190, (2, 20), generator=g)
    L = 80                                                          # crosses the mem_len = 64 ring boundary
    for pen in (0.0, 1.5):
        want, w_sc = ref_group_beam_search(ref, prompt, L, num_beams=4, num_beam_groups=2, diversity_penalty=pen, return_scores=True)
        dec = XLDecoder(m.engine, 2 * 4, L)
        got, g_sc = group_beam_search(dec, prompt.to(dev), L, num_beams=4, num_beam_groups=2, diversity_penalty=pen,
                                      return_scores=True)
        got = got.cpu()
        print(f'group beam (penalty {pen}) scores: HIP {g_sc.tolist()} oracle {w_sc.tolist()}')
        assert got.shape == want.shape and torch.equal(got[:, :20], prompt)
        assert (g_sc - w_sc).abs().max().item() < 2e-2
        same = (got == want).all(1)
        for b in (~same).nonzero().flatten().tolist():
            lp = ref(got[b:b + 1, :-1]).prediction_scores[0]
            s_got = lp[torch.arange(19, L - 1), got[b, 20:]].sum().item() / L
            assert abs(s_got - w_sc[b].item()) < 2e-2, (b, s_got, w_sc[b].item())
    # through the public API: all four beams returned; with a penalty the two groups do not emit the same first token
    out = m.generate(input_ids=prompt.to(dev), max_length=40, num_beams=4, num_beam_groups=2, diversity_penalty=5.0,
                     num_return_sequences=4, early_stopping=True).cpu()
    assert out.shape == (8, 40) and torch.equal(out[:, :20], prompt.repeat_interleave(4, 0))
    for b in range(2):
        assert len({int(t) for t in out[4 * b:4 * b + 4, 20]}) >= 2
    with pytest.raises(ValueError, match='divisible'):
        m.generate(input_ids=prompt.to(dev), max_length=40, num_beams=3, num_beam_groups=2)
    with pytest.raises(ValueError, match='sampling mode'):
        m.generate(input_ids=prompt.to(dev), max_length=40, num_beams=4, num_beam_groups=2, do_sample=True)


def test_contrastive_search_matches_oracle(dev):
    """The reference's 'contrastive' strategy (musicnlp/trainer/eval.py:296-302; its prepare_inputs_for_generation re-stacks the
    mems "to work with cosine sim generation", musicnlp/models/transformer_xl.py:229-234) against the oracle's restatement of
    HF 4.25.1 contrastive_search: token for token; a fork is accepted only at a step where the oracle's two best contrastive
    scores are within bf16 noise of each other.  alpha -> 0 reduces to greedy decoding."""
    from oracle.transfoxl_ref import ref_contrastive_search
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=160, seed=13)
    ref.eval(); m.eval()
    g = torch.Generator().manual_seed(16)
    prompt = torch.randint(4, 1190, (3, 24), generator=g)
    L = 100
    want, trace = ref_contrastive_search(ref, prompt, L, top_k=4, penalty_alpha=0.6, return_trace=True)
    got = m.generate(input_ids=prompt.to(dev), max_length=L, penalty_alpha=0.6, top_k=4).cpu()
    assert got.shape == want.shape and torch.equal(got[:, :24], prompt)
    greedy = ref.greedy_generate(prompt, max_length=L)
    assert (want != greedy).any(), 'the degeneration penalty must change something for the test to have power'
    for b in range(3):
        mism = (got[b] != want[b]).nonzero().flatten()
        if mism.numel():
            t = int(mism[0]) - 24
            sc = trace[t]['score'][b].sort(descending=True).values
            assert (sc[0] - sc[1]).item() < 2e-2, f'row {b} forks at step {t} where the oracle margin is {(sc[0] - sc[1]).item():.4f}'
    # after a legitimate fork the two continuations are different sequences: compare up to the first fork of each row
    prefix = [int((got[b] != want[b]).nonzero().flatten()[0]) - 24 if (got[b] != want[b]).any() else L - 24 for b in range(3)]
    print(f'contrastive search: generated tokens identical to the oracle before the first near-tie fork: {prefix} of {L - 24}')
    assert max(prefix) >= 20 and sum(prefix) >= 40
    tiny = m.generate(input_ids=prompt.to(dev), max_length=60, penalty_alpha=1e-9, top_k=4).cpu()
    gr = m.generate(input_ids=prompt.to(dev), max_length=60, do_sample=False).cpu()
    assert torch.equal(tiny, gr)


def test_contrastive_select_kernel_matches_formula(dev):
    """mxl_contrastive_select against HF's `_ranking_fast` formula in fp32 torch on the same bf16 inputs: scores to 1e-3, the
    same winner wherever the two best scores differ by more than that"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(3)
    B, K, S, Smax, d = 5, 6, 333, 400, 768
    ctx = torch.randn(B, Smax, d, device=dev).to(torch.bfloat16)
    hid = (ctx[:, 17:17 + K].float() * 0.7 + 0.5 * torch.randn(B, K, d, device=dev)).to(torch.bfloat16).reshape(B * K, d).contiguous()
    probs = torch.softmax(torch.randn(B, K, device=dev), -1).contiguous()
    inv = torch.empty(B, Smax, device=dev)
    for b in range(B):
        ops.row_inv_norm(ctx[b, :S], inv[b, :S], S)
    score = torch.empty(B * K, device=dev); sel = torch.empty(B, dtype=torch.int64, device=dev)
    ops.contrastive_select(ctx, inv, S, hid, probs, 0.6, score, sel)
    c = ctx[:, :S].float(); h = hid.float().view(B, K, d)
    cos = torch.einsum('bsd,bkd->bks', c / c.norm(dim=-1, keepdim=True), h / h.norm(dim=-1, keepdim=True))
    want = 0.4 * probs - 0.6 * cos.max(-1).values
    assert (score.view(B, K) - want).abs().max().item() < 1e-3
    top2 = want.topk(2, -1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2e-3
    assert torch.equal(sel[clear], want.argmax(-1)[clear]) and (inv[:, :S] - 1 / c.norm(dim=-1)).abs().max().item() < 1e-4


def test_two_lane_decoder_equals_single_decoder(dev):
    """generate.XLDecoderLanes (two free-running half-batch decoders, one hipGraph and one stream each): greedy tokens identical
    to the single decoder's, row for row; model.generate picks it from 32 rows on; under sampling eager == hipGraph replay"""
    from symbolic_music_generation_amd.generate import XLDecoder, XLDecoderLanes
    ref, m = _pair(dev, n_layer=2, mem_len=64, max_length=128, seed=21)
    m.eval()
    g = torch.Generator().manual_seed(22)
    prompt = torch.randint(4, 1190, (32, 16), generator=g).to(dev)
    one = XLDecoder(m.engine, 32, 128, seed=3).generate(prompt, 100, do_sample=False, use_graph=True)
    two = XLDecoderLanes(m.engine, 32, 128, seed=3, lanes=2).generate(prompt, 100, do_sample=False, use_graph=True)
    assert torch.equal(one, two)
    three = XLDecoderLanes(m.engine, 32, 128, seed=3, lanes=3)          # uneven lanes: 11 + 11 + 10 rows
    assert three.sizes == [11, 11, 10] and torch.equal(three.generate(prompt, 100, do_sample=False, use_graph=True), one)
    out = m.generate(input_ids=prompt, max_length=100, do_sample=False)
    assert type(m._decoder).__name__ == 'XLDecoderLanes' and torch.equal(out, one)
    kw = dict(max_length=90, do_sample=True, top_k=8, top_p=0.9)
    a = XLDecoderLanes(m.engine, 32, 128, seed=5, lanes=2).generate(prompt, use_graph=True, **kw)
    b = XLDecoderLanes(m.engine, 32, 128, seed=5, lanes=2).generate(prompt, use_graph=False, **kw)
    assert torch.equal(a, b)
    assert torch.equal(a[:, :16], prompt) and (a >= 0).all() and (a < 1190).all()
