"""Pins oracle/reformer_ref.py against fixtures produced by the real HuggingFace Reformer (tests/golden/make_reformer_goldens.py)."""
import os

import pytest
import torch

from oracle.reformer_ref import RefReformerConfig, RefReformer, auto_num_buckets, param_shapes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return torch.load(os.path.join(G, f'reformer_{name}.pt'), map_location='cpu', weights_only=False)


def _ref_from(blob):
    cfg = RefReformerConfig(**blob['config'])
    sd = {k: v for k, v in blob['state_dict'].items() if k in param_shapes(cfg)}
    assert set(sd) == set(param_shapes(cfg)), set(param_shapes(cfg)) ^ set(sd)
    m = RefReformer(cfg, sd)
    m.num_buckets = blob['num_buckets']
    return cfg, m


@pytest.mark.parametrize('name', ['single_chunk', 'chunked_h1', 'chunked_h2', 'dh64_h1'])
def test_oracle_matches_hf(name):
    blob = _load(name)
    cfg, m = _ref_from(blob)
    logits, loss = m.forward(blob['ids'], rotations=blob['rotations'], labels=blob['labels'])
    for l, b in blob['buckets'].items():
        assert torch.equal(m.last_buckets[l].to(torch.int32), b), f'bucket ids differ in layer {l}'
    assert (logits - blob['logits']).abs().max().item() < 2e-4
    assert abs(loss.item() - blob['loss'].item()) < 1e-5


def test_known_answers():
    # notebook/train/reformer.ipynb: 82.5 M parameters for base @ V=420, axial 64x64; num_buckets 128 at T=4096
    c = RefReformerConfig.from_preset('base', vocab_size=420, max_position_embeddings=4096, axial_pos_shape=(64, 64))
    n = sum(torch.Size(s).numel() for s in param_shapes(c).values())
    assert n == 82_498_980
    assert auto_num_buckets(4096, 64, 4096) == 128
    assert auto_num_buckets(8192, 64, 8192) == [16, 16]


@pytest.mark.parametrize('name', ['gen_short', 'gen_padded', 'gen_chunks'])
def test_cached_decoding_matches_hf(name):
    """the oracle's restatement of HF's cached decoding (ReformerDynamicCache: hidden states + bucket ids per layer, one token
    per step) against token ids and per-step logits recorded from the real HF implementation driven the way transformers 4.25.1's
    generate drove it: prompt shorter than a chunk (standard attention, then the first hashing once 64 positions exist), a
    padded prefill (pad bucket, widened offsets afterwards) and whole chunks"""
    blob = _load(name)
    cfg, m = _ref_from(blob)
    ids, logits = m.greedy_generate(blob['prompt'], blob['ids'].shape[1], blob['rotations'], return_logits=True)
    assert torch.equal(ids, blob['ids'])
    assert (logits - blob['step_logits']).abs().max().item() < 1e-4
