"""Incremental (cached) autoregressive decoding for the Reformer engine: `model.generate(...)` as the reference calls it
(musicnlp/trainer/eval.py:333) runs HF's `use_cache` path -- the prompt in one forward, then ONE token per forward against
`ReformerDynamicCache` (HF515:65-148).  Restated in oracle/reformer_ref.py (`prefill` / `step`, pinned on HF-recorded
fixtures); this is the device form:

    prompt pass   engine.forward on the prompt (padded to a multiple of 64 as HF's eval mode does: pads hash to an extra
                  bucket), every layer's projections and bucket ids copied into the caches
    one step      embed(token, position t) -> per layer [LN, projection (weight-streaming skinny GEMM), cache append,
                  single-query attention over <= 128 cached positions, output projection, LN, FFN] -> LN_2d -> head -> sampler
                  -- O(1) launches per token instead of a whole forward over the tokens so far

    local layer   keys = positions ((n // 64) - 1) * 64 .. t                                   (HF515:1327-1329)
    LSH layer     fewer than 64 bucketed positions: plain attention over all of them (self mask only); at 64 every cached
                  vector is hashed; from then on: hash the query, stable-sort (cached ; new) bucket ids per hash round, attend
                  to the 64-slot chunk holding the new token and the chunk before it           (HF515:482-534, 946-1050)

The caches hold projections ((B, Tmax, H*dh) bf16 per layer: k, v or shared qk, v), not hidden states: HF re-projects the
hidden states it gathers, which yields the same vectors.  The hash rotations are fixed for the whole generation (HF with
`config.hash_seed` set; with hash_seed=None, the reference's setting, HF redraws them every forward, which makes the cached
bucket ids of earlier tokens meaningless to later queries -- any fixed draw is a sample of that procedure's first step).
Batch rows are decoded independently (HF 5.15's cached LSH step gathers row 0's states for every row: HF515:1004-1006)."""
import math
from typing import Dict, Optional

import torch

from . import ops
from ._lib import MusicXLError


class RFDecoder:
    def __init__(self, engine, batch: int, max_total_len: int, rotations: Optional[Dict[int, torch.Tensor]] = None,
                 seed: int = 77):
        self.eng = engine
        c = engine.cfg
        self.B, self.Tmax = batch, max_total_len
        if batch > 64:
            raise MusicXLError('cached Reformer decoding: batch <= 64 (weight-streaming skinny GEMMs)')
        if max_total_len > c.max_position_embeddings:
            raise ValueError('max_length exceeds max_position_embeddings')
        dev = engine.dev
        d, H, dh, Fi = c.hidden_size, c.num_attention_heads, c.attention_head_size, c.feed_forward_size
        self.n_h = c.num_hashes
        if H * batch * self.n_h > 1024:
            raise MusicXLError('cached Reformer decoding: batch * heads * num_hashes <= 1024')
        bf = dict(device=dev, dtype=torch.bfloat16)
        L = len(c.attn_layers)
        self.kc = [torch.zeros(batch, max_total_len, d, **bf) for _ in range(L)]       # local: k        LSH: shared qk
        self.vc = [torch.zeros(batch, max_total_len, d, **bf) for _ in range(L)]
        self.bk = {l: torch.zeros(batch * H, self.n_h, max_total_len, device=dev, dtype=torch.int32)
                   for l, kind in enumerate(c.attn_layers) if kind == 'lsh'}
        self.bkmax = {l: torch.zeros(1, device=dev, dtype=torch.int32) for l in self.bk}
        self.n_bucketed = {l: 0 for l in self.bk}                                        # cached positions that have bucket ids
        self.ids = torch.zeros(batch, max_total_len + 1, device=dev, dtype=torch.int64)
        self.t_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.rng = torch.zeros(1, device=dev, dtype=torch.int64)
        self.seed = seed
        self.rotations = rotations
        self.x1 = torch.empty(batch, d, **bf); self.x2 = torch.empty(batch, d, **bf)
        self.y1 = torch.empty(batch, d, **bf); self.y2 = torch.empty(batch, d, **bf)
        self.hn = torch.empty(batch, d, **bf); self.h2 = torch.empty(batch, d, **bf)
        self.qkv = torch.empty(batch, 3 * d, **bf)
        self.av = torch.empty(batch, d, **bf); self.tmp = torch.empty(batch, d, **bf)
        self.a = torch.empty(batch, Fi, **bf)
        self.cat = torch.empty(batch, 2 * d, **bf); self.hcat = torch.empty(batch, 2 * d, **bf)
        self.logits = torch.empty(batch, engine.layout.head_rows_padded, device=dev, dtype=torch.float32)
        self.raw_bk = torch.empty(batch * H, self.n_h, device=dev, dtype=torch.int32)
        self.trace = None                   # optional (B, Tmax, V) f32: row t = logits computed FROM position t (parity tests)

    # ---------------------------------------------------------------- hashing helpers
    def _factors(self, T_hint: Optional[int] = None):
        """bucket factors.  A trained model carries `num_buckets` in its config (HF writes it back at the first chunked forward);
        on a fresh one HF derives it from the length of that forward -- here the padded prompt, or, for prompts within one
        chunk (where HF's cached path would fail on the unset value), the generation buffer"""
        e = self.eng
        if e.num_buckets is None:
            from .rf_engine import auto_num_buckets
            T = T_hint if T_hint is not None else max((self.Tmax + 63) // 64 * 64, 128)
            e.num_buckets = auto_num_buckets(T, 64, e.cfg.max_position_embeddings)
        nb = e.num_buckets
        return [nb] if isinstance(nb, int) else list(nb)

    def _rot(self, l):
        if self.rotations is None:
            self.rotations = {}
        if l not in self.rotations:
            c = self.eng.cfg
            g = torch.Generator(device=self.eng.dev).manual_seed((self.seed * 131 + l) & 0x7FFFFFFFFFFFFFFF)
            self.rotations[l] = torch.randn(c.num_attention_heads, c.attention_head_size, self.n_h, sum(self._factors()) // 2,
                                            device=self.eng.dev, generator=g)
        r = self.rotations[l]
        if r.device != self.eng.dev or r.dtype != torch.float32 or not r.is_contiguous():
            r = self.rotations[l] = r.to(self.eng.dev, torch.float32).contiguous()
        return r

    # ---------------------------------------------------------------- prompt
    def prefill(self, prompt: torch.Tensor, sampling: dict):
        e, c = self.eng, self.eng.cfg
        B, Tp = prompt.shape
        assert B == self.B and 1 <= Tp <= self.Tmax
        d, H = c.hidden_size, c.num_attention_heads
        self.ids.zero_()
        self.ids[:, :Tp].copy_(prompt)
        for l in self.bk:
            self.n_bucketed[l] = 0
            self.bkmax[l].zero_()
        Tf = Tp if Tp <= 64 else (Tp + 63) // 64 * 64          # HF pads beyond one chunk to a multiple of the chunk length
        pad = getattr(c, 'pad_token_id', None)
        buf = torch.full((B, Tf), 0 if pad is None else int(pad), device=e.dev, dtype=torch.int64)
        buf[:, :Tp].copy_(prompt)
        if Tf > 64:
            self._factors(Tf)
        rot = {l: self._rot(l) for l in self.bk} if Tf > 64 else None

        def sink(l, kind, qkv, buckets):
            rows = qkv.view(B, Tf, -1)[:, :Tp]
            if kind == 'local':
                self.kc[l][:, :Tp].copy_(rows[..., d:2 * d])
                self.vc[l][:, :Tp].copy_(rows[..., 2 * d:3 * d])
            else:
                self.kc[l][:, :Tp].copy_(rows[..., :d])
                self.vc[l][:, :Tp].copy_(rows[..., d:2 * d])
                if buckets is not None:
                    self.bk[l][:, :, :Tp].copy_(buckets.view(B * H, self.n_h, Tf)[:, :, :Tp])
                    self.bkmax[l].copy_(self.bk[l][:, :, :Tp].max().view(1))
                    self.n_bucketed[l] = Tp

        out = e.forward(buf, labels=None, train=False, rotations=rot, n_real=Tp if Tf > Tp else None, layer_sink=sink)
        last = out['logits'][:, Tp - 1].contiguous()
        self.t_dev.fill_(Tp - 1)
        self._sample(last, sampling)
        return out

    def _sample(self, logits, sampling):
        if self.trace is not None:
            self.trace.index_copy_(1, self.t_dev.to(torch.int64), logits[:, :self.trace.shape[-1]].unsqueeze(1))
        self._last = logits
        if sampling is None:                 # the caller picks the token (beam search)
            return
        ops.sample(logits[:, :self.eng.cfg.vocab_size], self.ids, self.t_dev, self.rng, self.seed, **sampling)
        ops.decode_advance(self.t_dev, self.rng)

    # ---------------------------------------------------------------- one token
    def step(self, t: int, sampling: dict):
        """consume the token at position t (already in `ids`), sample position t + 1"""
        e, c = self.eng, self.eng.cfg
        B, d, H, dh, Fi = self.B, c.hidden_size, c.num_attention_heads, c.attention_head_size, c.feed_forward_size
        L = len(c.attn_layers)
        A0, A1 = c.axial_pos_shape
        d0 = c.axial_pos_embds_dim[0]
        W0 = e.p32('reformer.embeddings.position_embeddings.weights.0').view(A0, d0)
        W1 = e.p32('reformer.embeddings.position_embeddings.weights.1').view(A1, d - d0)
        ops.rf_decode_embed(self.ids, t, e.w16('reformer.embeddings.word_embeddings.weight'), W0, W1, self.x1, A1)
        self.x2.copy_(self.x1)
        x1, x2, y1, y2 = self.x1, self.x2, self.y1, self.y2
        n = t + 1
        ops.ln_residual_fwd(x2, None, e._l(0, 'attention.layer_norm.weight', e.P), e._l(0, 'attention.layer_norm.bias', e.P),
                            self.hn, eps=c.layer_norm_eps)
        for l, kind in enumerate(c.attn_layers):
            nproj = 3 if kind == 'local' else 2
            qkv = self.qkv[:, :nproj * d]
            ops.gemm_skinny(self.hn, e._proj_w(l, kind), qkv, B, nproj * d, d, ldc=self.qkv.stride(0))
            if kind == 'local':
                self.kc[l][:, t].copy_(qkv[:, d:2 * d])
                self.vc[l][:, t].copy_(qkv[:, 2 * d:3 * d])
                start = max(((t // 64) - 1) * 64, 0)                      # HF515:1327-1329 (a negative start = all of them)
                ops.rf_decode_attn(qkv, self.kc[l], self.vc[l], None, self.av, B, H, dh, 1, self.Tmax, n, t, start=start,
                                   count=n - start, lsh=False)
            else:
                self.kc[l][:, t].copy_(qkv[:, :d])
                self.vc[l][:, t].copy_(qkv[:, d:2 * d])
                factors = self._factors()
                NB = math.prod(factors)
                if self.n_bucketed[l] == 0:
                    # no bucket ids yet (fewer than 64 positions when last looked at): plain attention over all n positions
                    ops.rf_decode_attn(qkv, self.kc[l], self.vc[l], None, self.av, B, H, dh, 1, self.Tmax, n, t, start=0, count=n,
                                       lsh=True)
                    if n >= 64:                                           # HF515:532-534: now every cached vector is hashed
                        tmpb = torch.empty(B * H, self.n_h * n, device=e.dev, dtype=torch.int32)
                        ops.lsh_hash(self.kc[l], self.Tmax * d, d, self._rot(l), tmpb, B, n, H, dh, self.n_h, factors)
                        self.bk[l][:, :, :n].copy_(tmpb.view(B * H, self.n_h, n))
                        self.bkmax[l].copy_(tmpb.max().view(1))
                        self.n_bucketed[l] = n
                else:
                    assert self.n_bucketed[l] == t
                    ops.lsh_hash(qkv, self.qkv.stride(0), self.qkv.stride(0), self._rot(l), self.raw_bk, B, 1, H, dh, self.n_h, factors)
                    ops.rf_query_bucket(self.raw_bk, self.bk[l], self.bkmax[l], B * H, self.n_h, NB, self.Tmax, t)
                    self.n_bucketed[l] = n
                    rows = self.bk[l][:, :, :n].contiguous().view(B * H * self.n_h, n)
                    sidx = torch.empty_like(rows)
                    spos = torch.empty_like(rows)
                    ops.lsh_sort(rows, sidx, spos, B * H * self.n_h, n, n, (NB + 1) * self.n_h)
                    ops.rf_decode_attn(qkv, self.kc[l], self.vc[l], sidx, self.av, B, H, dh, self.n_h, self.Tmax, n, t, lsh=True)
            # y1 = x1 + av Wo^T ; h2 = LN_ff(y1)
            ops.gemm_skinny(self.av, e._l(l, 'attention.output.dense.weight'), self.tmp, B, d, d)
            ops.ln_residual_fwd(self.tmp, x1, e._l(l, 'feed_forward.layer_norm.weight', e.P),
                                e._l(l, 'feed_forward.layer_norm.bias', e.P), self.h2, z=y1, eps=c.layer_norm_eps)
            # y2 = x2 + W2 relu(W1 h2 + b1) + b2 ; the next layer's attention LayerNorm rides on the same launch
            ops.gemm_skinny(self.h2, e._l(l, 'feed_forward.dense.dense.weight'), self.a, B, Fi, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU,
                            bias=e._l(l, 'feed_forward.dense.dense.bias', e.P))
            ops.gemm_skinny(self.a, e._l(l, 'feed_forward.output.dense.weight'), self.tmp, B, d, Fi, flags=ops.GEMM_BIAS,
                            bias=e._l(l, 'feed_forward.output.dense.bias', e.P))
            if l + 1 < L:
                ops.ln_residual_fwd(self.tmp, x2, e._l(l + 1, 'attention.layer_norm.weight', e.P),
                                    e._l(l + 1, 'attention.layer_norm.bias', e.P), self.hn, z=y2, eps=c.layer_norm_eps)
            else:
                ops.ln_residual_fwd(self.tmp, x2, e._l(l, 'feed_forward.layer_norm.weight', e.P),
                                    e._l(l, 'feed_forward.layer_norm.bias', e.P), self.hn, z=y2, eps=c.layer_norm_eps)   # only z is used
            x1, y1 = y1, x1
            x2, y2 = y2, x2
        self.cat[:, :d].copy_(x1)
        self.cat[:, d:].copy_(x2)
        ops.ln_residual_fwd(self.cat, None, e.p32('reformer.encoder.layer_norm.weight'), e.p32('reformer.encoder.layer_norm.bias'),
                            self.hcat, eps=c.layer_norm_eps)
        off = e.layout.entries['lm_head.decoder.weight'][0]
        nrow_p = e.layout.head_rows_padded
        head_w = e.W[off:off + nrow_p * 2 * d].view(nrow_p, 2 * d)
        V = c.vocab_size
        ops.gemm_skinny(self.hcat, head_w, self.logits, B, V, 2 * d, flags=ops.GEMM_OUT_F32 | ops.GEMM_BIAS, bias=e.p32('lm_head.bias'))
        self._sample(self.logits, sampling)

    # ---------------------------------------------------------------- beam-search hooks (generate.beam_search)
    def beam_prefill(self, prompt: torch.Tensor):
        self.prefill(prompt, None)

    def beam_logp(self) -> torch.Tensor:
        return torch.log_softmax(self._last[:, :self.eng.cfg.vocab_size].float(), -1)

    def beam_reorder(self, beam_idx: torch.Tensor):
        """rows follow their beams: id history, the projection caches and the bucket ids of every cached position"""
        H = self.eng.cfg.num_attention_heads
        self.ids.copy_(self.ids.index_select(0, beam_idx))
        for cache in self.kc + self.vc:
            cache.copy_(cache.index_select(0, beam_idx))
        for l, bk in self.bk.items():
            v = bk.view(self.B, H, self.n_h, self.Tmax)
            v.copy_(v.index_select(0, beam_idx))

    def beam_advance(self, cur_len: int):
        self.t_dev.fill_(cur_len - 1)
        self.step(cur_len - 1, None)

    # ---------------------------------------------------------------- loop
    @torch.no_grad()
    def generate(self, prompt: torch.Tensor, max_length: int, do_sample: bool = False, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, temperature: float = 1.0, repetition_penalty: Optional[float] = None,
                 typical_p: Optional[float] = None) -> torch.Tensor:
        if max_length > self.Tmax:
            raise MusicXLError(f'max_length {max_length} exceeds the decoder buffer {self.Tmax}')
        sampling = dict(do_sample=do_sample, top_k=top_k or 0, top_p=top_p if top_p is not None else 1.0, temperature=temperature,
                        repetition_penalty=repetition_penalty, typical_p=typical_p)
        Tp = prompt.shape[1]
        if max_length <= Tp:
            return prompt[:, :max_length]
        self.prefill(prompt.to(self.eng.dev), sampling)
        for t in range(Tp, max_length - 1):
            self.step(t, sampling)
        return self.ids[:, :max_length].clone()
