"""
ORACLE (test infrastructure, NOT product code) -- CPU fp32 restatement of the Reformer path of
StefanHeng/Symbolic-Music-Generation: `MyReformerModelWithLMHead` (musicnlp/models/reformer.py:90-127) is a pure
pass-through to HuggingFace `ReformerModelWithLMHead`, whose arithmetic is restated here in its own structure
(line numbers `HF515:` refer to transformers 5.15 modeling_reformer.py, the importable copy).

PINNED: `tests/golden/make_reformer_goldens.py` runs the real HF implementation in the build container (hash_seed set) and
stores weights / ids / logits / loss / bucket ids / rotations; `tests/test_reformer_oracle_cpu.py` checks this restatement
against those fixtures.  Version drift vs the reference's pinned transformers==4.25.1: 5.15's LM head never
adds `lm_head.bias` (zero at init, so init-time parity is unaffected); this restatement adds it, like 4.25.1.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this file.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

PRESETS = {  # musicnlp/models/reformer.py:15-44
    'debug': dict(max_position_embeddings=64, axial_pos_shape=(8, 8), hidden_size=128, num_attention_heads=8, n_pairs=3),
    'debug-large': dict(max_position_embeddings=512, axial_pos_shape=(16, 32), hidden_size=128, num_attention_heads=8, n_pairs=3),
    'tiny': dict(max_position_embeddings=1024, axial_pos_shape=(32, 32), hidden_size=256, num_attention_heads=8, n_pairs=3),
    'small': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=512, num_attention_heads=8, n_pairs=3),
    'base': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=768, num_attention_heads=12, n_pairs=6, num_hashes=2),
    'large': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=1024, num_attention_heads=16, n_pairs=12, num_hashes=2),
}


@dataclass
class RefReformerConfig:
    vocab_size: int = 420
    hidden_size: int = 768
    num_attention_heads: int = 12
    attention_head_size: int = 64
    feed_forward_size: int = 3072
    attn_layers: List[str] = field(default_factory=lambda: ['local', 'lsh'] * 6)
    max_position_embeddings: int = 2048
    axial_pos_shape: Tuple[int, int] = (32, 64)
    axial_pos_embds_dim: Tuple[int, int] = (192, 576)
    num_hashes: int = 1
    num_buckets: Union[None, int, List[int]] = None
    chunk_length: int = 64          # lsh_attn_chunk_length == local_attn_chunk_length == 64 (HF defaults, logged config)
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.05
    local_attention_probs_dropout_prob: float = 0.05
    lsh_attention_probs_dropout_prob: float = 0.0
    axial_norm_std: float = 1.0
    initializer_range: float = 0.02
    eos_token_id: int = 3
    pad_token_id: int = 1

    @staticmethod
    def from_preset(model_size='base', vocab_size=None, **kwargs) -> 'RefReformerConfig':
        p = dict(PRESETS[model_size])
        hd, nh = p['hidden_size'], p['num_attention_heads']
        assert hd % nh == 0 and hd % 4 == 0
        n_pairs = p.pop('n_pairs')
        p.update(attn_layers=['local', 'lsh'] * n_pairs, feed_forward_size=hd * 4, attention_head_size=hd // nh,
                 axial_pos_embds_dim=(hd // 4, 3 * hd // 4))
        if vocab_size is not None:
            p['vocab_size'] = vocab_size
        p.update(kwargs)
        c = RefReformerConfig(**p)
        assert len(c.axial_pos_shape) == 2 and c.axial_pos_shape[0] * c.axial_pos_shape[1] == c.max_position_embeddings
        return c


def auto_num_buckets(seq_len: int, chunk: int, max_pos: int):
    """HF515:791-809"""
    p2 = (2 * (seq_len // chunk)).bit_length() - 1
    nb = 2 ** p2
    limit = 2 * max(int((max_pos // chunk) ** 0.5), chunk)
    if nb > limit:
        return [2 ** (p2 // 2), 2 ** (p2 - p2 // 2)]
    return nb


def param_shapes(c: RefReformerConfig) -> "Dict[str, Tuple[int, ...]]":
    d, Fi, V = c.hidden_size, c.feed_forward_size, c.vocab_size
    hd = c.num_attention_heads * c.attention_head_size
    sh = {'reformer.embeddings.word_embeddings.weight': (V, d),
          'reformer.embeddings.position_embeddings.weights.0': (c.axial_pos_shape[0], 1, c.axial_pos_embds_dim[0]),
          'reformer.embeddings.position_embeddings.weights.1': (1, c.axial_pos_shape[1], c.axial_pos_embds_dim[1])}
    for i, kind in enumerate(c.attn_layers):
        p = f'reformer.encoder.layers.{i}.'
        sh[p + 'attention.layer_norm.weight'] = (d,)
        sh[p + 'attention.layer_norm.bias'] = (d,)
        if kind == 'local':
            for n in ('query', 'key', 'value'):
                sh[p + f'attention.self_attention.{n}.weight'] = (hd, d)
        else:
            for n in ('query_key', 'value'):
                sh[p + f'attention.self_attention.{n}.weight'] = (hd, d)
        sh[p + 'attention.output.dense.weight'] = (d, hd)
        sh[p + 'feed_forward.layer_norm.weight'] = (d,)
        sh[p + 'feed_forward.layer_norm.bias'] = (d,)
        sh[p + 'feed_forward.dense.dense.weight'] = (Fi, d)
        sh[p + 'feed_forward.dense.dense.bias'] = (Fi,)
        sh[p + 'feed_forward.output.dense.weight'] = (d, Fi)
        sh[p + 'feed_forward.output.dense.bias'] = (d,)
    sh['reformer.encoder.layer_norm.weight'] = (2 * d,)
    sh['reformer.encoder.layer_norm.bias'] = (2 * d,)
    sh['lm_head.decoder.weight'] = (V, 2 * d)
    sh['lm_head.bias'] = (V,)
    return sh


def init_params(c: RefReformerConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """HF `_init_weights`: Linear/Embedding N(0, initializer_range), biases 0, LayerNorm (1, 0), axial N(0, axial_norm_std)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n, s in param_shapes(c).items():
        if 'position_embeddings.weights' in n:
            out[n] = torch.randn(s, generator=g) * c.axial_norm_std
        elif n.endswith('layer_norm.weight'):
            out[n] = torch.ones(s)
        elif n.endswith('bias'):
            out[n] = torch.zeros(s)
        else:
            out[n] = torch.randn(s, generator=g) * c.initializer_range
    return out


def _look_back(x, n_chunks_dim=2):
    """keys/values of chunk c = [chunk c-1 (circular), chunk c]   (HF515:362-383 with num_chunks_before=1, after=0)"""
    prev = torch.cat([x[:, :, -1:], x[:, :, :-1]], dim=n_chunks_dim)
    return torch.cat([prev, x], dim=3)


def lsh_buckets(qk, rotations, num_buckets, pad_mask=None, increase_num_buckets=False):
    """qk (B,H,T,dh), rotations (H, dh, n_h, rot/2) -> offset bucket ids (B,H,n_h*T)   (HF515:698-770).  `pad_mask` (B,T) bool,
    True = real token: when any token is padding, pads go to ONE extra bucket and the per-round offsets use num_buckets + 1
    (HF515:746-756); `increase_num_buckets` widens the offsets the same way for a query hashed against such a cache."""
    rot = torch.einsum('bmtd,mdhr->bmhtr', qk.detach(), rotations)
    n_h = rotations.shape[2]
    if isinstance(num_buckets, int):
        r = torch.cat([rot, -rot], dim=-1)
        buckets = torch.argmax(r, dim=-1)
        nb = num_buckets
    else:
        buckets, cur_sum, cur_prod = None, 0, 1
        for f in num_buckets:
            rf = rot[..., cur_sum:cur_sum + f // 2]
            cur_sum += f // 2
            a = torch.argmax(torch.cat([rf, -rf], dim=-1), dim=-1)
            buckets = a if buckets is None else buckets + cur_prod * a
            cur_prod *= f
        nb = cur_prod
    if pad_mask is not None and not bool(pad_mask.all()):
        nb += 1
        buckets = torch.where(pad_mask[:, None, None, :].expand_as(buckets), buckets, torch.tensor(nb - 1))
    elif increase_num_buckets:
        nb += 1
    offsets = (torch.arange(n_h) * nb).view(1, 1, -1, 1)
    return (buckets + offsets).flatten(2, 3)


def chunked_attention(q, k, v, pos, chunk, self_mask: bool, drop_p=0.0):
    """q,k,v (B,H,S,dh) already in (sorted) slot order, pos (B,H,S) original positions.  Returns out (B,H,S,dh), lse (B,H,S).
    Causal mask on original positions with -1e9; optional self mask -1e5 applied after (HF515:852-890, 1257-1262)."""
    B, H, S, dh = q.shape
    nc = S // chunk
    qc = q.view(B, H, nc, chunk, dh)
    kc = _look_back(k.view(B, H, nc, chunk, dh))
    vc = _look_back(v.view(B, H, nc, chunk, dh))
    qp = pos.view(B, H, nc, chunk)
    kp = _look_back(qp.unsqueeze(-1)).squeeze(-1)
    dots = torch.matmul(qc, kc.transpose(-1, -2))
    causal = qp.unsqueeze(-1) >= kp.unsqueeze(-2)
    dots = torch.where(causal, dots, torch.tensor(-1e9))
    if self_mask:
        ne = qp.unsqueeze(-1) != kp.unsqueeze(-2)
        dots = torch.where(ne, dots, torch.tensor(-1e5))
    lse = torch.logsumexp(dots, dim=-1, keepdim=True)
    probs = torch.exp(dots - lse)
    probs = F.dropout(probs, p=drop_p, training=drop_p > 0)
    out = torch.matmul(probs, vc)
    return out.reshape(B, H, S, dh), lse.reshape(B, H, S)


class RefReformer:
    """Functional model over a parameter dict with HF state-dict names."""

    def __init__(self, c: RefReformerConfig, params: Dict[str, torch.Tensor]):
        self.c, self.p = c, params
        self.num_buckets = c.num_buckets
        self.last_buckets: Dict[int, torch.Tensor] = {}

    def _split(self, x):
        B, T, _ = x.shape
        return x.view(B, T, self.c.num_attention_heads, self.c.attention_head_size).transpose(1, 2)

    def _merge(self, x):
        B, H, T, dh = x.shape
        return x.transpose(1, 2).reshape(B, T, H * dh)

    def embed(self, ids):
        c, p = self.c, self.p
        B, T = ids.shape
        w0, w1 = p['reformer.embeddings.position_embeddings.weights.0'], p['reformer.embeddings.position_embeddings.weights.1']
        A0, A1 = c.axial_pos_shape
        pos = torch.cat([w0.expand(A0, A1, -1), w1.expand(A0, A1, -1)], dim=-1).reshape(A0 * A1, -1)[:T]
        return p['reformer.embeddings.word_embeddings.weight'][ids] + pos

    def local_attn(self, l, h):
        c, p = self.c, self.p
        pre = f'reformer.encoder.layers.{l}.attention.self_attention.'
        q = self._split(h @ p[pre + 'query.weight'].t())
        k = self._split(h @ p[pre + 'key.weight'].t()) / math.sqrt(c.attention_head_size)
        v = self._split(h @ p[pre + 'value.weight'].t())
        B, H, T, dh = q.shape
        pos = torch.arange(T).view(1, 1, T).expand(B, H, T)
        if T <= c.chunk_length:
            dots = q @ k.transpose(-1, -2)
            dots = torch.where(pos.unsqueeze(-1) >= pos.unsqueeze(-2), dots, torch.tensor(-1e9))
            out = torch.softmax(dots, -1) @ v
        else:
            out, _ = chunked_attention(q, k, v, pos, c.chunk_length, self_mask=False)
        return self._merge(out)

    def lsh_attn(self, l, h, rotations, buckets=None, pad_mask=None):
        c, p = self.c, self.p
        pre = f'reformer.encoder.layers.{l}.attention.self_attention.'
        qk = self._split(h @ p[pre + 'query_key.weight'].t())
        v = self._split(h @ p[pre + 'value.weight'].t())
        B, H, T, dh = qk.shape
        key = qk * torch.rsqrt(torch.mean(qk ** 2, -1, keepdim=True) + 1e-6) / math.sqrt(dh)   # HF515:1052-1066
        if T <= c.chunk_length:
            pos = torch.arange(T).view(1, 1, T).expand(B, H, T)
            dots = qk @ key.transpose(-1, -2)
            dots = torch.where(pos.unsqueeze(-1) >= pos.unsqueeze(-2), dots, torch.tensor(-1e9))
            dots = torch.where(pos.unsqueeze(-1) != pos.unsqueeze(-2), dots, torch.tensor(-1e5))
            return self._merge(torch.softmax(dots, -1) @ v)
        if self.num_buckets is None:
            self.num_buckets = auto_num_buckets(T, c.chunk_length, c.max_position_embeddings)
        if buckets is None:
            buckets = lsh_buckets(qk, rotations, self.num_buckets, pad_mask)         # (B,H,n_h*T)
        else:            # HF's LSHSelfAttention.forward takes ready-made `buckets` too (HF515:466-476): hashing is skipped
            buckets = buckets.view(B, H, -1).long()
        n_h = buckets.shape[-1] // T
        self.last_buckets[l] = buckets
        S = n_h * T
        scaled = S * buckets + (torch.arange(S).view(1, 1, -1) % S)                  # HF515:151-157
        sidx = torch.argsort(scaled, dim=-1)
        spos = sidx % T
        gather = spos.unsqueeze(-1).expand(-1, -1, -1, dh)
        qs, ks, vs = qk.gather(2, gather), key.gather(2, gather), v.gather(2, gather)
        out_s, lse_s = chunked_attention(qs, ks, vs, spos, c.chunk_length, self_mask=True)
        undo = torch.empty_like(sidx)
        undo.scatter_(-1, sidx, torch.arange(S).view(1, 1, -1).expand_as(sidx))
        out = out_s.gather(2, undo.unsqueeze(-1).expand(-1, -1, -1, dh)).view(B, H, n_h, T, dh)
        lse = lse_s.gather(2, undo).view(B, H, n_h, T, 1)
        if n_h > 1:
            w = torch.exp(lse - torch.logsumexp(lse, dim=2, keepdim=True))
            out = (out * w).sum(2)
        else:
            out = out[:, :, 0]
        return self._merge(out)

    def rotations_shape(self, T: int):
        c = self.c
        nb = self.num_buckets if self.num_buckets is not None else auto_num_buckets(T, c.chunk_length, c.max_position_embeddings)
        rot = nb if isinstance(nb, int) else sum(nb)
        return (c.num_attention_heads, c.attention_head_size, c.num_hashes, rot // 2)

    def forward(self, ids, rotations: Optional[Dict[int, torch.Tensor]] = None, labels=None,
                buckets: Optional[Dict[int, torch.Tensor]] = None):
        """ids (B,T); rotations: {layer index -> (H, dh, n_h, rot/2)} for every LSH layer (explicit input: HF draws them
        from the global RNG inside each layer, HF515:723-731).  Eval-mode semantics (no dropout)."""
        c, p = self.c, self.p
        d = c.hidden_size
        x = self.embed(ids)
        x1, x2 = x, x
        for l, kind in enumerate(c.attn_layers):
            pre = f'reformer.encoder.layers.{l}.'
            h = F.layer_norm(x2, (d,), p[pre + 'attention.layer_norm.weight'], p[pre + 'attention.layer_norm.bias'], c.layer_norm_eps)
            a = self.local_attn(l, h) if kind == 'local' else self.lsh_attn(l, h, (rotations or {}).get(l), (buckets or {}).get(l))
            y1 = x1 + a @ p[pre + 'attention.output.dense.weight'].t()
            h2 = F.layer_norm(y1, (d,), p[pre + 'feed_forward.layer_norm.weight'], p[pre + 'feed_forward.layer_norm.bias'], c.layer_norm_eps)
            f = torch.relu(h2 @ p[pre + 'feed_forward.dense.dense.weight'].t() + p[pre + 'feed_forward.dense.dense.bias'])
            y2 = x2 + f @ p[pre + 'feed_forward.output.dense.weight'].t() + p[pre + 'feed_forward.output.dense.bias']
            x1, x2 = y1, y2
        hcat = F.layer_norm(torch.cat([x1, x2], -1), (2 * d,), p['reformer.encoder.layer_norm.weight'],
                            p['reformer.encoder.layer_norm.bias'], c.layer_norm_eps)
        logits = hcat @ p['lm_head.decoder.weight'].t() + p['lm_head.bias']
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[:, :-1].reshape(-1, c.vocab_size), labels[:, 1:].reshape(-1), ignore_index=-100)
        return logits, loss


# ---------------------------------------------------------------------------------------------------------------------------
# Cached (incremental) decoding: `model.generate(...)` as musicnlp/trainer/eval.py:333 drives it runs HF's `use_cache` path --
# ReformerDynamicCache (HF515:65-148): per layer the LayerNorm'ed attention inputs of every position, plus, for LSH layers, the
# offset bucket ids (B, H, n_h, n).  A step feeds ONE token:
#   local layer (HF515:1136-1169, 1327-1329): keys / values = positions ((n // 64) - 1) * 64 .. n of the cache + the new token, no mask
#   LSH layer, fewer than 64 cached buckets (HF515:513-517, 840-845): standard attention of the query over all positions, self mask
#     only; once 64 positions exist ALL of them are hashed and the buckets enter the cache (HF515:532-534)
#   LSH layer with cached buckets (HF515:482-511, 946-1050): hash the query, stable-sort (cached buckets ; new bucket) per hash
#     round, take the 64-slot chunk that holds the new token and the chunk before it (indices modulo the list length), attend to
#     the hidden states at those positions (projected afresh; no causal mask -- every one of them is in the past -- self mask
#     -1e5 on the token itself), combine the rounds by their logsumexp weights.
# Rotations are an explicit input here ({layer: (H, dh, n_h, rot/2)}), i.e. HF with `config.hash_seed` set: the same rotations at
# every call.  (With hash_seed=None, the reference's setting, HF redraws them at every forward.)
# ---------------------------------------------------------------------------------------------------------------------------
class RefReformerCache:
    def __init__(self):
        self.states: List[torch.Tensor] = []          # per layer (B, n, d): LayerNorm(x2) of every position so far
        self.buckets: List[Optional[torch.Tensor]] = []   # per layer (B, H, n_h, n) or None


def _stable_argsort(v):
    n = v.shape[-1]
    return torch.argsort(n * v + (torch.arange(n) % n).view(*([1] * (v.dim() - 1)), n), dim=-1)   # HF515:151-157


def _rf_prefill(self, ids, rotations):
    """whole prompt with use_cache (HF515:2012-2040, 1380-1425): padded to a multiple of 64 when longer than one chunk (pads
    hash to an extra bucket); returns logits of the real positions and the cache"""
    c, p = self.c, self.p
    d = c.hidden_size
    B, Tp = ids.shape
    pad_mask = None
    if Tp > c.chunk_length and Tp % c.chunk_length:
        padn = c.chunk_length - Tp % c.chunk_length
        ids = torch.cat([ids, torch.full((B, padn), c.pad_token_id, dtype=ids.dtype)], 1)
        pad_mask = torch.arange(Tp + padn).view(1, -1).expand(B, -1) < Tp
    cache = RefReformerCache()
    x = self.embed(ids)
    x1, x2 = x, x
    for l, kind in enumerate(c.attn_layers):
        pre = f'reformer.encoder.layers.{l}.'
        h = F.layer_norm(x2, (d,), p[pre + 'attention.layer_norm.weight'], p[pre + 'attention.layer_norm.bias'], c.layer_norm_eps)
        if kind == 'local':
            a = self.local_attn(l, h)
            cache.buckets.append(None)
        else:
            a = self.lsh_attn(l, h, (rotations or {}).get(l), pad_mask=pad_mask)
            bk = None
            if ids.shape[1] > c.chunk_length:
                H = c.num_attention_heads
                bk = self.last_buckets[l].view(B, H, -1, ids.shape[1])[..., :Tp].clone()
            cache.buckets.append(bk)
        cache.states.append(h[:, :Tp])
        y1 = x1 + a @ p[pre + 'attention.output.dense.weight'].t()
        h2 = F.layer_norm(y1, (d,), p[pre + 'feed_forward.layer_norm.weight'], p[pre + 'feed_forward.layer_norm.bias'], c.layer_norm_eps)
        f = torch.relu(h2 @ p[pre + 'feed_forward.dense.dense.weight'].t() + p[pre + 'feed_forward.dense.dense.bias'])
        x1, x2 = y1, x2 + f @ p[pre + 'feed_forward.output.dense.weight'].t() + p[pre + 'feed_forward.output.dense.bias']
    hcat = F.layer_norm(torch.cat([x1, x2], -1), (2 * d,), p['reformer.encoder.layer_norm.weight'],
                        p['reformer.encoder.layer_norm.bias'], c.layer_norm_eps)
    logits = hcat @ p['lm_head.decoder.weight'].t() + p['lm_head.bias']
    return logits[:, :Tp], cache


def _rf_step(self, tok, cache, rotations):
    """one token (B, 1) against the cache -> logits (B, V); the cache grows by one position"""
    c, p = self.c, self.p
    d, H, dh, ch = c.hidden_size, c.num_attention_heads, c.attention_head_size, c.chunk_length
    B = tok.shape[0]
    t = cache.states[0].shape[1]                                   # position of the new token
    A0, A1 = c.axial_pos_shape
    w0, w1 = p['reformer.embeddings.position_embeddings.weights.0'], p['reformer.embeddings.position_embeddings.weights.1']
    pos = torch.cat([w0[t // A1, 0], w1[0, t % A1]], -1)
    x = p['reformer.embeddings.word_embeddings.weight'][tok] + pos   # (B, 1, d)
    x1, x2 = x, x
    for l, kind in enumerate(c.attn_layers):
        pre = f'reformer.encoder.layers.{l}.'
        sa = pre + 'attention.self_attention.'
        h = F.layer_norm(x2, (d,), p[pre + 'attention.layer_norm.weight'], p[pre + 'attention.layer_norm.bias'], c.layer_norm_eps)
        past = cache.states[l]
        if kind == 'local':
            start = ((past.shape[1] // ch) - 1) * ch                # HF515:1327-1329 (a negative start slices from the end)
            kv = torch.cat([past[:, start:], h], 1)
            q = self._split(h @ p[sa + 'query.weight'].t())
            k = self._split(kv @ p[sa + 'key.weight'].t()) / math.sqrt(dh)
            v = self._split(kv @ p[sa + 'value.weight'].t())
            a = self._merge(torch.softmax(q @ k.transpose(-1, -2), -1) @ v)
        else:
            rot = rotations[l]
            n_h = rot.shape[2]
            q = self._split(h @ p[sa + 'query_key.weight'].t())    # (B, H, 1, dh)
            allh = torch.cat([past, h], 1)                          # (B, t+1, d)
            if cache.buckets[l] is None:
                qk = self._split(allh @ p[sa + 'query_key.weight'].t())
                v = self._split(allh @ p[sa + 'value.weight'].t())
                key = qk * torch.rsqrt(torch.mean(qk ** 2, -1, keepdim=True) + 1e-6) / math.sqrt(dh)
                dots = q @ key.transpose(-1, -2)                    # (B, H, 1, t+1)
                dots[..., -1] = -1e5                                # self mask on the token itself (HF515:840-845)
                a = self._merge(torch.softmax(dots, -1) @ v)
                if t + 1 >= ch:                                     # HF515:532-534: from now on the buckets are cached
                    if self.num_buckets is None:
                        raise RuntimeError('num_buckets must be set before cached decoding hashes (HF sets it in the first chunked forward)')
                    cache.buckets[l] = lsh_buckets(qk, rot, self.num_buckets).view(B, H, n_h, t + 1)
            else:
                pb = cache.buckets[l]
                nbk = self.num_buckets if isinstance(self.num_buckets, int) else math.prod(self.num_buckets)
                inc = bool(pb.max() > n_h * nbk - 1)                # pad bucket was cached (HF515:961-965)
                qb = lsh_buckets(q, rot, self.num_buckets, increase_num_buckets=inc).view(B, H, n_h, 1)
                cb = torch.cat([pb, qb], -1)                        # (B, H, n_h, t+1)
                order = _stable_argsort(cb)
                n = t + 1
                rank = (order == n - 1).float().argmax(-1)          # sorted slot of the new token
                start = ((rank // ch) - 1) * ch
                slots = (start.unsqueeze(-1) + torch.arange(2 * ch)) % n
                posn = order.gather(-1, slots)                      # (B, H, n_h, 128) original positions
                hs = allh[torch.arange(B).view(B, 1, 1, 1), posn]   # (B, H, n_h, 128, d)
                wqk = p[sa + 'query_key.weight'].view(H, dh, d)
                wv = p[sa + 'value.weight'].view(H, dh, d)
                qk = torch.einsum('bhrld,hed->bhrle', hs, wqk)
                v = torch.einsum('bhrld,hed->bhrle', hs, wv)
                key = qk * torch.rsqrt(torch.mean(qk ** 2, -1, keepdim=True) + 1e-6) / math.sqrt(dh)
                dots = torch.einsum('bhe,bhrle->bhrl', q[:, :, 0], key)
                dots = torch.where(posn != n - 1, dots, torch.tensor(-1e5))
                lse = torch.logsumexp(dots, -1, keepdim=True)
                out = torch.einsum('bhrl,bhrle->bhre', torch.exp(dots - lse), v)
                if n_h > 1:
                    wgt = torch.exp(lse - torch.logsumexp(lse, 2, keepdim=True))
                    out = (out * wgt).sum(2)
                else:
                    out = out[:, :, 0]
                a = self._merge(out.unsqueeze(2))
                cache.buckets[l] = cb
        cache.states[l] = torch.cat([past, h], 1)
        y1 = x1 + a @ p[pre + 'attention.output.dense.weight'].t()
        h2 = F.layer_norm(y1, (d,), p[pre + 'feed_forward.layer_norm.weight'], p[pre + 'feed_forward.layer_norm.bias'], c.layer_norm_eps)
        f = torch.relu(h2 @ p[pre + 'feed_forward.dense.dense.weight'].t() + p[pre + 'feed_forward.dense.dense.bias'])
        x1, x2 = y1, x2 + f @ p[pre + 'feed_forward.output.dense.weight'].t() + p[pre + 'feed_forward.output.dense.bias']
    hcat = F.layer_norm(torch.cat([x1, x2], -1), (2 * d,), p['reformer.encoder.layer_norm.weight'],
                        p['reformer.encoder.layer_norm.bias'], c.layer_norm_eps)
    return (hcat @ p['lm_head.decoder.weight'].t() + p['lm_head.bias'])[:, 0]


@torch.no_grad()
def _rf_greedy_generate(self, ids, max_length, rotations, return_logits=False):
    """HF GenerationMixin.greedy_search over the cached path: arg-max of the last position's logits"""
    logits, cache = self.prefill(ids, rotations)
    last = logits[:, -1]
    trace = [last]
    while ids.shape[1] < max_length:
        nxt = last.argmax(-1, keepdim=True)
        ids = torch.cat([ids, nxt], 1)
        if ids.shape[1] == max_length:
            break
        last = self.step(nxt, cache, rotations)
        trace.append(last)
    return (ids, torch.stack(trace, 1)) if return_logits else ids


RefReformer.prefill = _rf_prefill
RefReformer.step = _rf_step
RefReformer.greedy_generate = _rf_greedy_generate
