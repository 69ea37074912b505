"""Round 6, the tail of a decode step (musicnlp/models/transformer_xl.py:223-241 is the step it serves): sampler + embedding row of
the sampled token + counter advance in one launch (mxl_sample_step), no log-softmax launch where nothing reads log-probabilities
-- against the launches it replaces, BIT-identical tokens, op level and through the whole decoder, eager and under hipGraph replay."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('sampling', [dict(do_sample=False), dict(do_sample=True, top_k=8), dict(do_sample=True, top_k=40, top_p=0.9, temperature=0.8),
                                      dict(do_sample=True, top_k=0, top_p=0.95, typical_p=0.9), dict(do_sample=True, top_k=8, repetition_penalty=1.3)])
def test_sample_step_equals_sampler_embedding_advance(dev, sampling):
    """mxl_sample_step == mxl_sample + mxl_decode_embed of the sampled token + mxl_decode_advance: same token, same embedding row,
    same counters; and (no repetition penalty) the head's raw logits give the same tokens as their log-softmax"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(3)
    B, V, d, T = 64, 1190, 768, 40
    logits = (torch.randn(B, 1216) * 3).to(dev)
    logp = torch.log_softmax(logits[:, :V], -1).contiguous()
    E = bf(torch.randn(V, d)).to(dev)
    ids0 = torch.randint(4, V, (B, T + 2), device=dev)
    scale = d ** 0.5

    def state(t):
        return ids0.clone(), torch.tensor([t], device=dev, dtype=torch.int32), torch.tensor([11], device=dev, dtype=torch.int64)

    ids_a, t_a, rng_a = state(T)
    ops.sample(logp, ids_a, t_a, rng_a, 99, **sampling)
    ops.decode_advance(t_a, rng_a)
    emb_a = torch.empty(B, d, device=dev, dtype=torch.bfloat16)
    ops.decode_embed(ids_a, t_a, E, emb_a, scale)
    ctr = torch.zeros(1, device=dev, dtype=torch.int32)
    for scores, raw in ((logp, False), (logits, True)):
        if raw and sampling.get('repetition_penalty', 1.0) != 1.0:
            continue
        ids_b, t_b, rng_b = state(T)
        emb_b = torch.full((B, d), float('nan'), device=dev, dtype=torch.bfloat16)
        ops.sample_step(scores, V, ids_b, t_b, rng_b, 99, E, emb_b, scale, ctr, **sampling)
        torch.cuda.synchronize()
        assert torch.equal(ids_a, ids_b), f'raw logits: {raw}'
        assert torch.equal(emb_a, emb_b) and int(t_b.item()) == T + 1 and int(rng_b.item()) == 12 and int(ctr.item()) == 0


def _model(dev, seed, dh=64, **kw):
    from tests.test_xl_model_gpu import _pair
    return _pair(dev, n_layer=3, mem_len=64, max_length=160, seed=seed, n_head=128 // dh, d_head=dh, **kw)


@pytest.mark.parametrize('dh', [64, 16])
@pytest.mark.parametrize('sampling', [dict(do_sample=False), dict(do_sample=True, top_k=8), dict(do_sample=True, top_k=8, repetition_penalty=1.2)])
def test_fused_sampler_decoder_equals_five_launch_tail(dev, monkeypatch, sampling, dh):
    """the whole decoder with the one-launch step tail against the five-launch one: identical tokens over a generation that wraps
    the ring, eager and under hipGraph replay, with and without the log-softmax launch (trace on / off); identical log-probabilities
    where traced"""
    from symbolic_music_generation_amd.generate import XLDecoder
    ref, m = _model(dev, 41, dh=dh)
    m.eval()
    g = torch.Generator().manual_seed(42)
    prompt = torch.randint(4, 1190, (5, 12), generator=g).to(dev)
    V = 1190

    def run(unfused, use_graph, trace):
        monkeypatch.setenv('MXL_DECODE_UNFUSED', '1' if unfused else '0')
        dec = XLDecoder(m.engine, 5, 160, seed=9)
        assert dec.fused_sampler == (not unfused)
        if trace:
            dec.trace = torch.zeros(5, 161, V, device=dev)
        out = dec.generate(prompt, 150, use_graph=use_graph, **sampling)
        return out, dec.trace

    long_ids, long_tr = run(True, False, True)
    for use_graph in (False, True):
        for trace in (True, False):
            ids, tr = run(False, use_graph, trace)
            assert torch.equal(ids, long_ids), (use_graph, trace)
            if trace:
                assert torch.equal(tr, long_tr)


def test_fused_sampler_greedy_token_parity_with_oracle(dev):
    """the one-launch step tail against the CPU oracle's HF-style loop (prompt, then one token at a time with carried mems), dh = 64"""
    ref, m = _model(dev, 43)
    ref.eval(); m.eval()
    prompt = torch.randint(4, 1190, (3, 24))
    want = ref.greedy_generate(prompt, max_length=120)
    got = m.generate(input_ids=prompt.to(dev), max_length=120, do_sample=False, use_graph=True).cpu()
    mism = (got != want).nonzero()
    assert mism.numel() == 0, f'first divergence at {mism[0].tolist()}'
