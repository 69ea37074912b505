"""Generates tests/golden/reformer_*.pt in the BUILD container from the real HuggingFace Reformer (transformers 5.15, the
importable copy of the implementation behind musicnlp/models/reformer.py).  Stored: config values, state_dict, input ids,
logits, loss, per-LSH-layer bucket ids and the rotation tensors HF drew (re-drawn here with the same seed/shape).

    python tests/golden/make_reformer_goldens.py
"""
import os
import sys

import torch
from transformers import ReformerConfig, ReformerModelWithLMHead
from transformers.models.reformer import modeling_reformer as mr

HERE = os.path.dirname(os.path.abspath(__file__))
HASH_SEED = 1234


def make(name, T, hidden, heads, axial, n_pairs, num_hashes, vocab=97, seed=0):
    torch.manual_seed(seed)
    cfg = ReformerConfig(
        attn_layers=['local', 'lsh'] * n_pairs, hidden_size=hidden, num_attention_heads=heads,
        attention_head_size=hidden // heads, feed_forward_size=4 * hidden, max_position_embeddings=axial[0] * axial[1],
        axial_pos_shape=axial, axial_pos_embds_dim=(hidden // 4, 3 * hidden // 4), is_decoder=True, num_buckets=None,
        num_hashes=num_hashes, vocab_size=vocab, hash_seed=HASH_SEED, eos_token_id=3, pad_token_id=1,
        # reference / HF defaults in force: chunk 64/64, 1 chunk before, relu, eps 1e-12, dropouts 0.05/0.05/0.0
    )
    model = ReformerModelWithLMHead(cfg).eval()
    with torch.no_grad():   # non-trivial scales so hashing / softmax are exercised
        for n, p in model.named_parameters():
            if p.dim() > 1 and 'position_embeddings' not in n:
                p.mul_(4.0)
            if 'layer_norm.weight' in n:
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith('bias') and 'lm_head' not in n:
                p.add_(0.05 * torch.randn_like(p))
    ids = torch.randint(4, vocab, (2, T))
    labels = ids.clone()
    labels[1, T - 9:] = -100
    buckets = {}
    orig = mr.LSHSelfAttention._hash_vectors

    def spy(self, vectors, num_hashes, attention_mask, increase_num_buckets=False):
        b = orig(self, vectors, num_hashes, attention_mask, increase_num_buckets)
        buckets[self.layer_idx] = b.clone()
        return b

    mr.LSHSelfAttention._hash_vectors = spy
    with torch.no_grad():
        out = model(input_ids=ids, labels=labels)
    mr.LSHSelfAttention._hash_vectors = orig
    rotations = {}
    nb = model.config.num_buckets
    if T > 64:
        rot = nb if isinstance(nb, int) else sum(nb)
        for l, kind in enumerate(cfg.attn_layers):
            if kind == 'lsh':
                torch.manual_seed(HASH_SEED)   # exactly what HF515:723-731 does
                rotations[l] = torch.randn(heads, hidden // heads, num_hashes, rot // 2)
    blob = dict(
        config=dict(vocab_size=vocab, hidden_size=hidden, num_attention_heads=heads, attention_head_size=hidden // heads,
                    feed_forward_size=4 * hidden, attn_layers=['local', 'lsh'] * n_pairs,
                    max_position_embeddings=axial[0] * axial[1], axial_pos_shape=tuple(axial),
                    axial_pos_embds_dim=(hidden // 4, 3 * hidden // 4), num_hashes=num_hashes),
        num_buckets=nb, state_dict={k: v.clone() for k, v in model.state_dict().items()},
        ids=ids, labels=labels, logits=out.logits.clone(), loss=out.loss.clone(),
        buckets={k: v.to(torch.int32) for k, v in buckets.items()}, rotations=rotations, hash_seed=HASH_SEED,
    )
    path = os.path.join(HERE, f'reformer_{name}.pt')
    torch.save(blob, path)
    print(name, 'T', T, 'num_buckets', nb, 'loss', out.loss.item(), 'bytes', os.path.getsize(path))


def make_generate(name, Tp, L, hidden=64, heads=2, axial=(16, 16), n_pairs=2, num_hashes=2, vocab=97, seed=0):
    """Greedy decoding through HF's cached path (`use_cache`, ReformerDynamicCache), driven exactly as GenerationMixin did in the
    reference's transformers 4.25.1: the prompt in one forward, then one token per forward with `past_buckets_states`.  (5.15's
    own `generate` no longer threads the Reformer cache and recomputes the whole sequence every step.)  One sequence, as the
    reference generates (musicnlp/trainer/eval.py:333) -- for batch rows > 0 HF's cached LSH step gathers the hidden states of
    row 0 (`batch_size * torch.div(offset, n)` at HF515:1004-1006 is always 0), so only B = 1 is a meaningful fixture."""
    torch.manual_seed(seed)
    cfg = ReformerConfig(
        attn_layers=['local', 'lsh'] * n_pairs, hidden_size=hidden, num_attention_heads=heads,
        attention_head_size=hidden // heads, feed_forward_size=4 * hidden, max_position_embeddings=axial[0] * axial[1],
        axial_pos_shape=axial, axial_pos_embds_dim=(hidden // 4, 3 * hidden // 4), is_decoder=True, num_buckets=None,
        num_hashes=num_hashes, vocab_size=vocab, hash_seed=HASH_SEED, eos_token_id=3, pad_token_id=1)
    model = ReformerModelWithLMHead(cfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() > 1 and 'position_embeddings' not in n:
                p.mul_(4.0)
        model(input_ids=torch.randint(4, vocab, (1, axial[0] * axial[1])))     # fixes config.num_buckets as a full-length forward does
    nb = model.config.num_buckets
    ids = torch.randint(4, vocab, (1, Tp))
    prompt = ids.clone()
    with torch.no_grad():
        out = model(input_ids=ids, use_cache=True)
        cache, trace = out.past_buckets_states, [out.logits[:, -1]]
        while ids.shape[1] < L:
            nxt = trace[-1].argmax(-1, keepdim=True)
            ids = torch.cat([ids, nxt], 1)
            if ids.shape[1] == L:
                break
            out = model(input_ids=nxt, past_buckets_states=cache, use_cache=True)
            cache = out.past_buckets_states
            trace.append(out.logits[:, -1])
    rot = nb if isinstance(nb, int) else sum(nb)
    rotations = {}
    for l, kind in enumerate(cfg.attn_layers):
        if kind == 'lsh':
            torch.manual_seed(HASH_SEED)
            rotations[l] = torch.randn(heads, hidden // heads, num_hashes, rot // 2)
    blob = dict(
        config=dict(vocab_size=vocab, hidden_size=hidden, num_attention_heads=heads, attention_head_size=hidden // heads,
                    feed_forward_size=4 * hidden, attn_layers=['local', 'lsh'] * n_pairs,
                    max_position_embeddings=axial[0] * axial[1], axial_pos_shape=tuple(axial),
                    axial_pos_embds_dim=(hidden // 4, 3 * hidden // 4), num_hashes=num_hashes, pad_token_id=1),
        num_buckets=nb, state_dict={k: v.clone() for k, v in model.state_dict().items()}, prompt=prompt, ids=ids,
        step_logits=torch.stack(trace, 1), rotations=rotations, hash_seed=HASH_SEED)
    path = os.path.join(HERE, f'reformer_{name}.pt')
    torch.save(blob, path)
    print(name, 'prompt', Tp, '->', L, 'num_buckets', nb, 'bytes', os.path.getsize(path))


if __name__ == '__main__':
    make_generate('gen_short', Tp=20, L=150)             # standard attention, then the first hashing at 64 cached positions
    make_generate('gen_padded', Tp=70, L=200)            # padded prefill: pad bucket, widened offsets in the cached steps
    make_generate('gen_chunks', Tp=128, L=256, n_pairs=1, num_hashes=1, seed=4)
    make('single_chunk', T=64, hidden=64, heads=2, axial=(8, 8), n_pairs=1, num_hashes=1)
    make('chunked_h1', T=256, hidden=64, heads=2, axial=(16, 16), n_pairs=1, num_hashes=1)
    make('chunked_h2', T=256, hidden=64, heads=2, axial=(16, 16), n_pairs=2, num_hashes=2)
    make('dh64_h1', T=512, hidden=128, heads=2, axial=(16, 32), n_pairs=1, num_hashes=1)
