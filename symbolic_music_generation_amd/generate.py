"""Autoregressive generation for the TransfoXL engine: the on-device counterpart of `model.generate(...)` as the
reference calls it (musicnlp/trainer/eval.py:277-333; HF 4.25.1 GenerationMixin.greedy_search / sample).

    prompt forward (whole prompt, zero mems)  ->  K/V rings filled
    loop:  [embed -> L x (qkv GEMM, kv append, ring attention, o GEMM, LN, FFN GEMMs, LN) -> head GEMM -> log-softmax
            -> sampler -> advance]            one hipGraph replay per token, no host round trip

Strategies mirrored from `MusicGenerator` (eval.py:277-326): greedy (do_sample=False) and sampling with
`top_k`, `top_p`, `typical_p`, `temperature`, `repetition_penalty` and renormalised logits -- every key the `sample`
strategy accepts (eval.py:279) -- and beam search (`strategy='beam'`, eval.py:302-321: HF `beam_search` / `beam_sample` with
`BeamSearchScorer`), which runs the same per-token kernels eagerly with the beam bookkeeping between steps (XLDecoder.beam_search).
Contrastive search raises: HF 4.25.1's `contrastive_search` requires `past_key_values` in the model output, which neither
TransfoXL (`mems`) nor Reformer (`past_buckets_states`) returns, so that strategy fails in the reference too.
"""
import math
import os
from typing import Optional

import torch

from . import ops
from ._lib import MusicXLError


class XLDecoder:
    def __init__(self, engine, batch: int, max_total_len: int, seed: int = 77):
        self.eng = engine
        c = engine.cfg
        self.B, self.Tmax = batch, max_total_len
        dev = engine.dev
        d, M, L, Fi = c.d_model, c.mem_len, c.n_layer, c.d_inner
        bf = dict(device=dev, dtype=torch.bfloat16)
        H, dh = c.n_head, c.d_head
        self.kc = [torch.zeros(batch, H, M, dh, **bf) for _ in range(L)]    # head-major rings
        self.vc = [torch.zeros(batch, H, M, dh, **bf) for _ in range(L)]
        self.rd = None                                  # per-layer Rd tables (eval: no dropout on pos_emb)
        self.ids = torch.zeros(batch, max_total_len + 1, device=dev, dtype=torch.int64)
        self.t_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.rng = torch.zeros(1, device=dev, dtype=torch.int64)
        self.seed = seed
        self.h = [torch.empty(batch, d, **bf) for _ in range(2)]
        self.qkv = torch.empty(batch, 3 * d, **bf)
        self.av = torch.empty(batch, d, **bf)
        self.qr = torch.empty(batch, d, **bf)
        self.slabs = torch.zeros(4, 64, d, device=dev, dtype=torch.float32)   # K-slice partials of the FFN output projection
        self.bd = torch.empty(batch, H, M, device=dev, dtype=torch.float32)
        # ring pieces per (sequence, head): the attention launch fills every CU with the same number of bytes (ops.decode_ring_pieces)
        self.pieces = ops.decode_ring_pieces(batch, H, M)
        self.split = ops.relattn_decode_split_scratch(batch, H, dh, self.pieces, dev)
        self.tmp = torch.empty(batch, d, **bf)
        self.h1 = torch.empty(batch, d, **bf)
        self.a = torch.empty(batch, Fi, **bf)
        self.logits = torch.empty(batch, engine.layout.head_rows_padded, device=dev, dtype=torch.float32)
        self.logp = torch.empty(batch, c.vocab_size, device=dev, dtype=torch.float32)
        self.graph = None
        self._graph_key = None
        # Round 6: sampler + the sampled token's embedding row (the next step's input) + counter advance in ONE launch
        # (mxl_sample_step), and no log-softmax launch where nothing reads log-probabilities: 2 launches per step around the layers
        # instead of 5.  MXL_DECODE_UNFUSED=1 keeps the five (A/B runs, tests).
        self.fused_sampler = c.vocab_size <= 2048 and os.environ.get('MXL_DECODE_UNFUSED') != '1'
        self.step_ctr = torch.zeros(1, device=dev, dtype=torch.int32)
        # optional (B, Tmax, V) f32 buffer: row t receives the log-probs computed FROM position t (parity tests compare them
        # with a one-shot forward); written on the device by position, so it also works under hipGraph replay
        self.trace = None

    def _tables(self):
        if self.rd is None:
            e, c = self.eng, self.eng.cfg
            phi = ops.sinusoid_table(c.mem_len, c.d_model, c.clamp_len, e.dev)
            self.rd = []
            for l in range(c.n_layer):
                rd = torch.empty(c.mem_len, c.d_model, device=e.dev, dtype=torch.bfloat16)
                ops.gemm(phi, e._lw(l, 'dec_attn.r_net.weight'), rd, c.mem_len, c.d_model, c.d_model)
                self.rd.append(rd)

    def invalidate_tables(self):
        """call after the weights change (r_net feeds the cached Rd tables)"""
        self.rd = None
        self.graph = None

    # ---------------------------------------------------------------- prompt
    def prefill(self, prompt: torch.Tensor, sampling: dict):
        """Whole prompt through the training-shape kernels with zero mems (upstream first step), rings filled from the
        per-layer qkv buffers, first new token sampled from the last position."""
        e, c = self.eng, self.eng.cfg
        B, Tp = prompt.shape
        assert B == self.B and Tp + 1 <= self.Tmax + 1
        self._tables()
        for k in self.kc + self.vc:
            k.zero_()
        self.ids.zero_()
        self.ids[:, :Tp].copy_(prompt)
        sink_kc, sink_vc = self.kc, self.vc

        def kv_sink(l, qkv):
            ops.kv_fill(qkv, sink_kc[l], sink_vc[l], Tp)

        out = e.forward(prompt.to(e.dev), mems=None, labels=None, train=False, want_logprobs=False, kv_sink=kv_sink)
        ws = e._last
        N = B * Tp
        # log-probs of the last prompt position only
        if ws.logits is None:       # bucketed (large-vocabulary) head: the (N, V) logits were never formed; project B rows here
            self.tmp.copy_(ws.hid.view(B, Tp, -1)[:, Tp - 1])
            self._head(self.tmp)
        else:
            last = ws.logits.view(B, Tp, -1)[:, Tp - 1]
            ops.adaptive_logprob(last, self.logp, B, c.vocab_size, tuple(c.cutoffs))
        self.t_dev.fill_(Tp - 1)
        self._trace()
        if sampling is not None:                       # None: the caller picks the token from self.logp (beam search)
            self._sample_advance(self.logp, sampling)  # t = Tp: position of the token just sampled
        return out

    def _sample_advance(self, scores, sampling: dict):
        """next token of every row from `scores` (log-probabilities, or the head's logits: see mxl_sample_step) -> ids[:, t + 1];
        position and RNG counters advanced.  Short chain: the same launch leaves the token's embedding row in h[0] for the next step."""
        c = self.eng.cfg
        if self.fused_sampler:
            ops.sample_step(scores, c.vocab_size, self.ids, self.t_dev, self.rng, self.seed,
                            self.eng.w16('transformer.word_emb.emb_layers.0.weight'), self.h[0], math.sqrt(c.d_model), self.step_ctr,
                            **sampling)
        else:
            ops.sample(scores[:, :c.vocab_size] if scores.shape[1] != c.vocab_size else scores, self.ids, self.t_dev, self.rng,
                       self.seed, **sampling)
            ops.decode_advance(self.t_dev, self.rng)

    # ---------------------------------------------------------------- one token
    def force_tokens(self, tokens: torch.Tensor):
        """teacher forcing / constrained decoding: `tokens` (B,) replace the ids at the current position t (what the sampler just
        wrote) before the next `step`; with the fused sampler the embedding row the sampler left for that step follows"""
        self.ids.index_copy_(1, self.t_dev.to(torch.int64), tokens.to(self.ids.device, torch.int64).unsqueeze(1))
        if self.fused_sampler:
            c = self.eng.cfg
            ops.decode_embed(self.ids, self.t_dev, self.eng.w16('transformer.word_emb.emb_layers.0.weight'), self.h[0], math.sqrt(c.d_model))

    def step(self, sampling: dict, want_logp: bool = False):
        """one more token for every row.  want_logp: self.logp is needed after the step (it always holds the log-probabilities when
        a trace is attached, the head is adaptive or a repetition penalty is in force)"""
        # the log-softmax launch is only needed for what reads log-probabilities: the trace, an adaptive (clustered) head, and the
        # repetition penalty (sign-dependent); every other warper and the draw itself are shift-invariant (mxl_sample_step)
        raw = (self.fused_sampler and self.trace is None and not want_logp and not tuple(self.eng.cfg.cutoffs)
               and float(sampling.get('repetition_penalty', 1.0) or 1.0) == 1.0)
        self._forward_token(embed=not self.fused_sampler, want_logp=not raw)
        self._sample_advance(self.logits if raw else self.logp, sampling)

    def _forward_token(self, embed: bool = True, want_logp: bool = True):
        """the token at position t (ids[:, t], t on the device) through the model: K/V appended to the rings at slot t mod M,
        self.logp = log-probabilities of position t + 1 (want_logp=False: only self.logits, the head's raw rows).
        embed=False: h[0] already holds the token's embedding row (written by the previous step's sampler launch)."""
        e, c = self.eng, self.eng.cfg
        B, d, H, dh, M, Fi, L = self.B, c.d_model, c.n_head, c.d_head, c.mem_len, c.d_inner, c.n_layer
        E = e.w16('transformer.word_emb.emb_layers.0.weight')
        G = ops.gemm_skinny if B <= 64 else ops.gemm     # weight-streaming form for decode batches
        if embed:
            ops.decode_embed(self.ids, self.t_dev, E, self.h[0], math.sqrt(d))
        for l in range(L):
            h_in, h_out = self.h[l & 1], self.h[(l + 1) & 1]
            rrb = e._lw(l, 'dec_attn.r_r_bias', e.P)
            if B <= 64:      # projection, ring append and q + r_r_bias in one launch
                ops.decode_qkv(h_in, e._lw(l, 'dec_attn.qkv_net.weight'), self.qkv, self.kc[l], self.vc[l], self.t_dev,
                               rrb.reshape(-1), self.qr, dh)
            else:
                G(h_in, e._lw(l, 'dec_attn.qkv_net.weight'), self.qkv, B, 3 * d, d)
                ops.kv_append(self.qkv, self.kc[l], self.vc[l], self.t_dev, rrb=rrb.reshape(-1), qr_out=self.qr)
            ops.relattn_decode(self.qkv, self.kc[l], self.vc[l], self.rd[l], e._lw(l, 'dec_attn.r_w_bias', e.P),
                               rrb, self.av, self.t_dev, H, dh, self.qr, self.bd, qr_ready=True, split=self.split, pieces=self.pieces)
            G(self.av, e._lw(l, 'dec_attn.o_net.weight'), self.tmp, B, d, d)
            ops.ln_residual_fwd(self.tmp, h_in, e._lw(l, 'dec_attn.layer_norm.weight', e.P),
                                e._lw(l, 'dec_attn.layer_norm.bias', e.P), self.h1, eps=c.layer_norm_epsilon)
            G(self.h1, e._lw(l, 'pos_ff.CoreNet.0.weight'), self.a, B, Fi, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU,
              bias=e._lw(l, 'pos_ff.CoreNet.0.bias', e.P))
            if B <= 64:
                # N = d columns are only d/16 workgroups: slice K four ways as well; the slabs are summed by the LayerNorm launch
                ops.gemm_skinny_partial(self.a, e._lw(l, 'pos_ff.CoreNet.3.weight'), self.slabs, B, d, Fi, 4)
                ops.ln_residual_fwd_partial(self.slabs, 4, e._lw(l, 'pos_ff.CoreNet.3.bias', e.P), self.h1,
                                            e._lw(l, 'pos_ff.layer_norm.weight', e.P), e._lw(l, 'pos_ff.layer_norm.bias', e.P),
                                            h_out, eps=c.layer_norm_epsilon)
            else:
                G(self.a, e._lw(l, 'pos_ff.CoreNet.3.weight'), self.tmp, B, d, Fi, flags=ops.GEMM_BIAS,
                  bias=e._lw(l, 'pos_ff.CoreNet.3.bias', e.P))
                ops.ln_residual_fwd(self.tmp, self.h1, e._lw(l, 'pos_ff.layer_norm.weight', e.P),
                                    e._lw(l, 'pos_ff.layer_norm.bias', e.P), h_out, eps=c.layer_norm_epsilon)
        self._head(self.h[L & 1], want_logp)
        if want_logp:
            self._trace()

    def _head(self, hid, want_logp: bool = True):
        """(B, d) hidden states -> self.logp: all head rows (vocabulary + cluster rows) in one weight-streaming GEMM, then the
        adaptive log-softmax over the row (HF `ProjectedAdaptiveLogSoftmax.log_prob`, the labels=None branch)"""
        e, c = self.eng, self.eng.cfg
        B, d = self.B, c.d_model
        G = ops.gemm_skinny if B <= 64 else ops.gemm
        nrow, nrow_p = e.layout.n_head_rows, e.layout.head_rows_padded
        head_w = e.W[:nrow_p * d].view(nrow_p, d)
        boff = e.layout.entries['crit.out_layers.0.bias'][0]
        G(hid, head_w, self.logits, B, nrow, d, flags=ops.GEMM_OUT_F32 | ops.GEMM_BIAS, bias=e.P[boff:boff + nrow])
        if want_logp:
            ops.adaptive_logprob(self.logits, self.logp, B, c.vocab_size, tuple(c.cutoffs))

    # ---------------------------------------------------------------- beam-search hooks (see beam_search below)
    def beam_prefill(self, prompt: torch.Tensor):
        self.prefill(prompt, None)

    def beam_logp(self) -> torch.Tensor:
        return self.logp

    def beam_reorder(self, beam_idx: torch.Tensor):
        """rows follow their beams: id history and both rings of every layer (HF `_reorder_cache`: index_select on the mems)"""
        self.ids.copy_(self.ids.index_select(0, beam_idx))
        for ring in self.kc + self.vc:
            ring.copy_(ring.index_select(0, beam_idx))

    def beam_advance(self, cur_len: int):
        """the token at position cur_len - 1 through the model -> log-probs of position cur_len"""
        self.t_dev.fill_(cur_len - 1)
        self._forward_token()

    def _trace(self):
        if self.trace is not None:
            self.trace.index_copy_(1, self.t_dev.to(torch.int64), self.logp.unsqueeze(1))

    # ---------------------------------------------------------------- loop
    def begin(self, prompt: torch.Tensor, max_length: int, sampling: dict, use_graph: bool = True) -> int:
        """prompt pass + first sampled token + (use_graph) capture of one decode step; returns the number of `replay_once()`
        calls that complete the generation to max_length"""
        if max_length > self.Tmax:
            raise MusicXLError(f'max_length {max_length} exceeds the decoder buffer {self.Tmax}')
        self._sampling = sampling
        self._use_graph = use_graph
        self.prefill(prompt, sampling)
        steps = max_length - prompt.shape[1] - 1
        if steps > 0 and use_graph:
            key = tuple(sorted(sampling.items()))
            if self.graph is None or self._graph_key != key:
                # warm-up on a side stream (first launches set function attributes), then capture one step
                state = (self.t_dev.clone(), self.rng.clone(), self.ids.clone(),
                         [k.clone() for k in self.kc], [v.clone() for v in self.vc], self.h[0].clone())
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.step(sampling)
                torch.cuda.current_stream().wait_stream(s)
                self.graph = torch.cuda.CUDAGraph()       # hipGraph on ROCm
                with torch.cuda.graph(self.graph):
                    self.step(sampling)
                self._graph_key = key
                # restore the state the two extra steps consumed
                self.t_dev.copy_(state[0]); self.rng.copy_(state[1]); self.ids.copy_(state[2])
                for a, b in zip(self.kc, state[3]):
                    a.copy_(b)
                for a, b in zip(self.vc, state[4]):
                    a.copy_(b)
                self.h[0].copy_(state[5])         # (short chain: the next step's embedding row is step state too)
        return max(steps, 0)

    def replay_once(self):
        """one more token for every row (on the current stream)"""
        if self._use_graph:
            self.graph.replay()
        else:
            self.step(self._sampling)

    def generate(self, prompt: torch.Tensor, max_length: int, do_sample: bool = False, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, temperature: float = 1.0, repetition_penalty: Optional[float] = None,
                 typical_p: Optional[float] = None, use_graph: bool = True) -> torch.Tensor:
        """Returns (B, max_length) ids = prompt + continuation.  Like the reference (eos_token_id stays HF's default 0 =
        [OMIT]) decoding runs to max_length."""
        sampling = dict(do_sample=do_sample, top_k=top_k or 0, top_p=top_p if top_p is not None else 1.0,
                        temperature=temperature, repetition_penalty=1.0 if repetition_penalty is None else repetition_penalty,
                        typical_p=1.0 if typical_p is None else typical_p)
        Tp = prompt.shape[1]
        if max_length - Tp <= 0:
            return prompt[:, :max_length]
        for _ in range(self.begin(prompt, max_length, sampling, use_graph)):
            self.replay_once()
        return self.ids[:, :max_length].clone()


class XLDecoderLanes:
    """The batch as `lanes` independent XLDecoders of B / lanes sequences, each with its own hipGraph, replayed on its own
    stream.  A decode step is ~100 small dependent launches around 12 ring-attention launches: alone, the small launches run at
    launch / latency cost with the chip idle and the ring streaming waits for them; with two lanes one lane's small launches
    overlap the other's ring streaming (sequences are independent: the lanes never synchronise until the generation ends).
    Same interface as XLDecoder for `generate` / `begin` / `replay_once`; rows keep their order."""

    def __init__(self, engine, batch: int, max_total_len: int, seed: int = 1234, lanes: int = 2):
        assert 1 <= lanes <= batch
        self.B, self.Tmax, self.n = batch, max_total_len, lanes
        self.sizes = [batch // lanes + (1 if i < batch % lanes else 0) for i in range(lanes)]
        self.offs = [sum(self.sizes[:i]) for i in range(lanes + 1)]
        # (the sampler draws per (seed, row, step): a different seed per lane keeps the lanes' draws independent)
        self.lanes = [XLDecoder(engine, b, max_total_len, seed=seed + 7919 * i) for i, b in enumerate(self.sizes)]
        self.streams = [torch.cuda.Stream() for _ in range(lanes)]

    def invalidate_tables(self):
        for d in self.lanes:
            d.invalidate_tables()

    def begin(self, prompt, max_length, sampling, use_graph=True) -> int:
        steps = [d.begin(prompt[self.offs[i]:self.offs[i + 1]], max_length, sampling, use_graph) for i, d in enumerate(self.lanes)]
        for s in self.streams:                       # the lanes start from the prompt passes and captures issued above
            s.wait_stream(torch.cuda.current_stream())
        return steps[0]

    def replay_once(self):
        for d, s in zip(self.lanes, self.streams):
            with torch.cuda.stream(s):
                d.replay_once()

    def join(self):
        for s in self.streams:
            torch.cuda.current_stream().wait_stream(s)

    def generate(self, prompt, max_length, do_sample=False, top_k=None, top_p=None, temperature=1.0, repetition_penalty=None,
                 typical_p=None, use_graph=True) -> torch.Tensor:
        sampling = dict(do_sample=do_sample, top_k=top_k or 0, top_p=top_p if top_p is not None else 1.0,
                        temperature=temperature, repetition_penalty=1.0 if repetition_penalty is None else repetition_penalty,
                        typical_p=1.0 if typical_p is None else typical_p)
        if max_length - prompt.shape[1] <= 0:
            return prompt[:, :max_length]
        for _ in range(self.begin(prompt, max_length, sampling, use_graph)):
            self.replay_once()
        self.join()
        return torch.cat([d.ids[:, :max_length] for d in self.lanes], 0)


class _BeamHyps:
    """HF 4.25.1 generation/beam_search.py `BeamHypotheses`: the best finished hypotheses of one batch item"""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping: bool):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams, self.worst_score = [], 1e9

    def add(self, hyp: torch.Tensor, sum_logprobs: float):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                ranked = sorted((sc, i) for i, (sc, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self.beams) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def beam_search(dec, prompt: torch.Tensor, max_length: int, num_beams: int = 3, do_sample: bool = False,
                top_k: Optional[int] = None, top_p: Optional[float] = None, temperature: float = 1.0,
                typical_p: Optional[float] = None, early_stopping: bool = True, length_penalty: float = 1.0,
                num_return_sequences: int = 1, eos_token_id: int = 0, pad_token_id: Optional[int] = None,
                renormalize_logits: bool = True, generator: Optional[torch.Generator] = None, return_scores: bool = False):
    """HF 4.25.1 `beam_search` (do_sample=False) / `beam_sample` (do_sample=True) with `BeamSearchScorer.process / finalize`,
    as `model.generate(num_beams=...)` reaches them from musicnlp/trainer/eval.py:302-333.  `dec` is an XLDecoder or an
    rf_generate.RFDecoder (anything with beam_prefill / beam_logp / beam_reorder / beam_advance and `ids`) with one row per beam:
    B * num_beams rows (times num_return_sequences for beam_sample, as HF expands).  Per step: the device computes the
    log-probabilities of every beam's next token (the same kernels as `step`), the 2 * num_beams best (or sampled) continuations
    per item are taken on the device, the scorer's walk over them runs on the host (it is a data-dependent loop over a handful
    of scalars, as in HF), and the K/V rings and the id history follow their beams (HF `_reorder_cache`).
    Returns (B * num_return_sequences, L) ids, padded with pad_token_id (= eos when the config has none, as HF does)."""
    e, c = dec.eng, dec.eng.cfg
    dev, V = e.dev, c.vocab_size
    nb = num_beams
    if nb < 2:
        raise MusicXLError('beam search needs num_beams > 1')
    pad = eos_token_id if pad_token_id is None else pad_token_id
    B0, Tp = prompt.shape
    Bs = B0 * (num_return_sequences if do_sample else 1)              # scorer batch (HF: batch_size * num_return_sequences)
    keep = 1 if do_sample else num_return_sequences
    if keep > nb:
        raise MusicXLError('num_return_sequences has to be smaller or equal to num_beams')
    rows = Bs * nb
    if dec.B != rows or max_length > dec.Tmax:
        raise MusicXLError(f'the decoder was built for {dec.B} rows x {dec.Tmax} positions, beam search needs {rows} x {max_length}')
    expanded = prompt.repeat_interleave(rows // B0, 0).to(dev)
    dec.beam_prefill(expanded)
    beam_scores = torch.zeros(Bs, nb, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(nb, length_penalty, early_stopping) for _ in range(Bs)]
    done = [False] * Bs
    cur_len = Tp
    ident = torch.arange(rows, device=dev)
    while True:
        sc = dec.beam_logp() + beam_scores[:, None]
        if do_sample:              # HF beam_sample: warp log p + beam score, renormalise, draw 2 * num_beams, sort
            sc = _warp(sc, top_k, top_p, typical_p, temperature, min_keep=2, renormalize=renormalize_logits)
            flat = sc.view(Bs, nb * V)
            pick = torch.multinomial(torch.softmax(flat, -1), 2 * nb, generator=generator)
            top_s, order = flat.gather(-1, pick).sort(descending=True, dim=1)
            top_i = pick.gather(-1, order)
        else:
            top_s, top_i = sc.view(Bs, nb * V).topk(2 * nb, dim=1, largest=True, sorted=True)
        top_b, top_t = (top_i // V).tolist(), (top_i % V).tolist()
        top_sl = top_s.tolist()
        n_s, n_t, n_i = [[0.0] * nb for _ in range(Bs)], [[pad] * nb for _ in range(Bs)], [[0] * nb for _ in range(Bs)]
        for b in range(Bs):
            if done[b]:
                continue
            k = 0
            for rank in range(2 * nb):
                tok, s_, src = top_t[b][rank], top_sl[b][rank], b * nb + top_b[b][rank]
                if tok == eos_token_id:
                    if rank >= nb:
                        continue
                    hyps[b].add(dec.ids[src, :cur_len].clone(), s_)
                else:
                    n_s[b][k], n_t[b][k], n_i[b][k] = s_, tok, src
                    k += 1
                if k == nb:
                    break
            if k < nb:
                raise MusicXLError(f'at most {nb} tokens in the top {2 * nb} can be eos')
            done[b] = done[b] or hyps[b].is_done(max(top_sl[b]), cur_len)
        beam_scores = torch.tensor(n_s, device=dev).view(-1)
        beam_idx = torch.tensor(n_i, device=dev).view(-1)
        if not torch.equal(beam_idx, ident):
            dec.beam_reorder(beam_idx)
        dec.ids[:, cur_len] = torch.tensor(n_t, device=dev).view(-1)
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
        dec.beam_advance(cur_len)
    final = beam_scores.tolist()
    for b in range(Bs):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(dec.ids[b * nb + j, :cur_len].clone(), final[b * nb + j])
    best, scores = [], []
    for b in range(Bs):
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(keep):
            sc_, h = ranked.pop()
            best.append(h); scores.append(sc_)
    L = min(max(len(h) for h in best) + 1, max_length)
    out = torch.full((len(best), L), pad, dtype=torch.int64, device=dev)
    for i, h in enumerate(best):
        out[i, :len(h)] = h
        if len(h) < L:
            out[i, len(h)] = eos_token_id
    return (out, torch.tensor(scores)) if return_scores else out


def group_beam_search(dec, prompt: torch.Tensor, max_length: int, num_beams: int = 4, num_beam_groups: int = 2,
                      diversity_penalty: float = 0.0, early_stopping: bool = True, length_penalty: float = 1.0,
                      num_return_sequences: int = 1, eos_token_id: int = 0, pad_token_id: Optional[int] = None,
                      return_scores: bool = False):
    """HF 4.25.1 `group_beam_search` (diverse beam search, Vijayakumar et al.) with `BeamSearchScorer(num_beam_groups=...)` and
    `HammingDiversityLogitsProcessor`, as `model.generate(num_beams=, num_beam_groups=, diversity_penalty=)` reaches them from
    the reference's 'beam' strategy (musicnlp/trainer/eval.py:303-317: num_beam_groups set => do_sample False).  One decoder row
    per beam; per step ONE forward for all beams, then the groups in order: group g's log-probabilities are lowered by
    diversity_penalty x (how many beams of the EARLIER groups of the same item chose that token at this step), its 2 x group_size
    best continuations go through the scorer walk (one hypothesis heap and one done flag per item, shared by its groups, as in
    4.25.1), and its rows follow their beams.  The first beam of every group starts at score 0, the others at -1e9."""
    e, c = dec.eng, dec.eng.cfg
    dev, V = e.dev, c.vocab_size
    nb, ng = num_beams, num_beam_groups
    if ng < 2 or nb % ng != 0:
        raise ValueError('`num_beams` should be divisible by `num_beam_groups` for group beam search.')      # HF's message
    gs = nb // ng
    pad = eos_token_id if pad_token_id is None else pad_token_id
    B0, Tp = prompt.shape
    if num_return_sequences > nb:
        raise MusicXLError('num_return_sequences has to be smaller or equal to num_beams')
    rows = B0 * nb
    if dec.B != rows or max_length > dec.Tmax:
        raise MusicXLError(f'the decoder was built for {dec.B} rows x {dec.Tmax} positions, group beam search needs {rows} x {max_length}')
    dec.beam_prefill(prompt.repeat_interleave(nb, 0).to(dev))
    beam_scores = torch.full((B0, nb), -1e9, device=dev)
    beam_scores[:, ::gs] = 0
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(nb, length_penalty, early_stopping) for _ in range(B0)]
    done = [False] * B0
    cur_len = Tp
    ident = torch.arange(rows, device=dev)
    while True:
        logp = dec.beam_logp()                                   # (rows, V), every beam of every group
        current = torch.zeros(rows, dtype=torch.int64, device=dev)
        reorder = ident.clone()
        new_scores = beam_scores.clone()
        for g in range(ng):
            g0 = g * gs
            gidx = (torch.arange(B0, device=dev)[:, None] * nb + g0 + torch.arange(gs, device=dev)[None, :]).view(-1)
            sc = logp.index_select(0, gidx)
            if diversity_penalty and diversity_penalty > 0.0 and g > 0:
                # HammingDiversityLogitsProcessor: tokens the earlier groups of the same item have just chosen
                prev = current.view(B0, nb)[:, :g0]
                freq = torch.zeros(B0, V, device=dev).scatter_add_(1, prev, torch.ones_like(prev, dtype=torch.float32))
                sc = sc - diversity_penalty * freq.repeat_interleave(gs, 0)
            sc = sc + beam_scores.index_select(0, gidx)[:, None]
            top_s, top_i = sc.view(B0, gs * V).topk(2 * gs, dim=1, largest=True, sorted=True)
            top_b, top_t, top_sl = (top_i // V).tolist(), (top_i % V).tolist(), top_s.tolist()
            n_s, n_t, n_i = [[0.0] * gs for _ in range(B0)], [[pad] * gs for _ in range(B0)], [[0] * gs for _ in range(B0)]
            for b in range(B0):
                if done[b]:
                    n_i[b] = [b * nb + g0 + j for j in range(gs)]
                    continue
                k = 0
                for rank in range(2 * gs):
                    tok, s_, src = top_t[b][rank], top_sl[b][rank], b * nb + g0 + top_b[b][rank]
                    if tok == eos_token_id:
                        if rank >= gs:
                            continue
                        hyps[b].add(dec.ids[src, :cur_len].clone(), s_)
                    else:
                        n_s[b][k], n_t[b][k], n_i[b][k] = s_, tok, src
                        k += 1
                    if k == gs:
                        break
                if k < gs:
                    raise MusicXLError(f'at most {gs} tokens in the top {2 * gs} can be eos')
                done[b] = done[b] or hyps[b].is_done(max(top_sl[b]), cur_len)
            new_scores[gidx] = torch.tensor(n_s, device=dev).view(-1)
            current[gidx] = torch.tensor(n_t, device=dev).view(-1)
            reorder[gidx] = torch.tensor(n_i, device=dev).view(-1)
        beam_scores = new_scores
        if not torch.equal(reorder, ident):
            dec.beam_reorder(reorder)
        dec.ids[:, cur_len] = current
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
        dec.beam_advance(cur_len)
    final = beam_scores.tolist()
    for b in range(B0):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(dec.ids[b * nb + j, :cur_len].clone(), final[b * nb + j])
    best, scores = [], []
    for b in range(B0):
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(num_return_sequences):
            sc_, h = ranked.pop()
            best.append(h); scores.append(sc_)
    L = min(max(len(h) for h in best) + 1, max_length)
    out = torch.full((len(best), L), pad, dtype=torch.int64, device=dev)
    for i, h in enumerate(best):
        out[i, :len(h)] = h
        if len(h) < L:
            out[i, len(h)] = eos_token_id
    return (out, torch.tensor(scores)) if return_scores else out


def contrastive_search(dec, prompt: torch.Tensor, max_length: int, top_k: int = 4, penalty_alpha: float = 0.6,
                       eos_token_id: Optional[int] = 0, pad_token_id: Optional[int] = None) -> torch.Tensor:
    """HF 4.25.1 `GenerationMixin.contrastive_search` over Transformer-XL mems -- the reference's 'contrastive' strategy
    (musicnlp/trainer/eval.py:296-302); the reference's `prepare_inputs_for_generation` re-stacks the per-row mems lists that
    routine builds ("to work with cosine sim generation", musicnlp/models/transformer_xl.py:229-234).  One decoder row per
    (sequence, candidate): the K = top_k rows of a sequence share its history.  Per step: (1) the top-k tokens of the current
    log-probabilities and their probabilities renormalised over those k (TopKLogitsWarper, then softmax); (2) all B * K
    candidates through one cached decode step; (3) mxl_contrastive_select scores each candidate (1 - alpha) * p - alpha * max
    cosine similarity between its last-layer hidden state and those of every earlier position, and picks the best; (4) the K
    rows of the sequence take over the picked candidate's rings and history, its log-probabilities open the next step."""
    e, c = dec.eng, dec.eng.cfg
    dev, d, L = e.dev, c.d_model, c.n_layer
    K = int(top_k)
    if K < 2 or not penalty_alpha or penalty_alpha <= 0:
        raise ValueError('contrastive search needs top_k > 1 and penalty_alpha > 0')
    B0, Tp = prompt.shape
    rows = B0 * K
    if dec.B != rows or max_length > dec.Tmax:
        raise MusicXLError(f'the decoder was built for {dec.B} rows x {dec.Tmax} positions, contrastive search needs {rows} x {max_length}')
    pad = eos_token_id if pad_token_id is None else pad_token_id
    dec.beam_prefill(prompt.repeat_interleave(K, 0).to(dev))
    ws = e._last
    hid0 = (ws.h[L & 1]).view(rows, Tp, d)[::K]                          # last-layer output of every prompt position
    ctx = torch.empty(B0, max_length, d, device=dev, dtype=torch.bfloat16)
    inv = torch.empty(B0, max_length, device=dev, dtype=torch.float32)
    ctx[:, :Tp].copy_(hid0)
    for b in range(B0):
        ops.row_inv_norm(ctx[b, :Tp], inv[b, :Tp], Tp)
    logp = dec.beam_logp()[::K].clone()                                  # (B0, V)
    score = torch.empty(rows, device=dev, dtype=torch.float32)
    sel = torch.empty(B0, device=dev, dtype=torch.int64)
    grp = torch.arange(B0, device=dev) * K
    unfinished = torch.ones(B0, dtype=torch.bool, device=dev)
    cur_len = Tp
    while cur_len < max_length:
        top_lp, top_ids = logp.topk(K, dim=-1)
        probs = torch.softmax(top_lp.float(), dim=-1).contiguous()
        dec.ids[:, cur_len] = top_ids.reshape(-1)
        dec.beam_advance(cur_len + 1)                                    # every candidate at position cur_len
        hid = dec.h[L & 1]
        ops.contrastive_select(ctx, inv, cur_len, hid, probs, penalty_alpha, score, sel)
        src = grp + sel
        if eos_token_id is not None:                                     # finished sequences emit pad from now on
            tok = dec.ids[src, cur_len]
            tok = torch.where(unfinished, tok, torch.full_like(tok, pad))
        dec.beam_reorder(src.repeat_interleave(K))
        if eos_token_id is not None:
            dec.ids[:, cur_len] = tok.repeat_interleave(K)
            unfinished = unfinished & (tok != eos_token_id)
        ctx[:, cur_len].copy_(hid.index_select(0, src))
        ops.row_inv_norm(ctx[:, cur_len], inv[:, cur_len], B0)
        logp = dec.beam_logp().index_select(0, src)
        cur_len += 1
        if eos_token_id is not None and not bool(unfinished.any()):
            break
    return dec.ids[::K, :cur_len].clone()


def _warp(scores: torch.Tensor, top_k, top_p, typical_p, temperature, min_keep: int, renormalize: bool = True) -> torch.Tensor:
    """HF 4.25.1 logits warpers in `_get_logits_warper` order (temperature, top-k, top-p, typical-p, then
    LogitNormalization when renormalize_logits is set -- the reference sets it for every sampling call, eval.py:323) on a
    (rows, V) score matrix; min_tokens_to_keep = 2 under beam search.  Used by beam_sample only -- plain sampling runs in the
    sampler kernel.  HF applies the warpers AFTER adding the running beam scores, so with LogitNormalization every beam's row is
    renormalised to log-sum-exp 0 each step and the running score drops out of the draw; that is the reference's behaviour and
    it is kept."""
    neg = float('-inf')
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k:
        k = min(max(top_k, min_keep), scores.shape[-1])
        scores = scores.masked_fill(scores < scores.topk(k, -1).values[..., -1:], neg)
    if top_p is not None and top_p < 1.0:
        srt, idx = scores.sort(descending=False, dim=-1)
        remove = srt.softmax(-1).cumsum(-1) <= (1 - top_p)
        remove[..., -min_keep:] = False
        scores = scores.masked_fill(remove.scatter(-1, idx, remove), neg)
    if typical_p is not None and typical_p < 1.0:
        logp = scores.log_softmax(-1)
        p = logp.exp()
        ent = -(torch.nan_to_num(logp * p, nan=0.0)).sum(-1, keepdim=True)
        shifted = ((-logp) - ent).abs()
        srt, idx = shifted.sort(descending=False, dim=-1)
        cum = scores.gather(-1, idx).softmax(-1).cumsum(-1)
        last = (cum < typical_p).sum(-1)
        last[last < 0] = 0
        remove = srt > srt.gather(-1, last.view(-1, 1).clamp(max=scores.shape[-1] - 1))
        if min_keep > 1:
            remove[..., :min_keep] = False
        scores = scores.masked_fill(remove.scatter(-1, idx, remove), neg)
    return scores.log_softmax(-1) if renormalize else scores


# -------------------------------------------------------------------- bar-aligned cuts around generation
def truncate_last_bar(ids: torch.Tensor, sob_token_id: int):
    """`MusicGenerator._truncate_last_bar` (musicnlp/trainer/eval.py:178-185) for a batch: every generated row cut just before
    its last start-of-bar token, so a bar broken off by `max_length` is not rendered.  ids: (T,) or (B, T) int64 on the GPU;
    returns a list of ints (1-D input, as the reference) or a list of such lists.  Like the reference it refuses a row with no
    start-of-bar token."""
    one = ids.dim() == 1
    x = ids.view(1, -1) if one else ids
    x = x.contiguous()
    cut = ops.find_token(x, sob_token_id, -1).tolist()
    if min(cut) < 0:
        raise MusicXLError('no start-of-bar token found in a sequence to truncate')
    host = x.cpu()
    rows = [host[b, :c].tolist() for b, c in enumerate(cut)]
    return rows[0] if one else rows


def truncate_first_n_bar(ids: torch.Tensor, sob_token_id: int, n_bar: int = 8) -> torch.Tensor:
    """`MusicGenerator.truncate_first_n_bar` (eval.py:187-198) on ids: the prefix of one song up to (not including) its
    start-of-bar number `n_bar` (0-based, i.e. the song header plus the first `n_bar` bars), with a start-of-bar appended as the
    prompt for generation.  ids: (T,) int64 on the GPU; returns a (n,) int64 device tensor."""
    assert ids.dim() == 1
    cut = int(ops.find_token(ids.view(1, -1).contiguous(), sob_token_id, n_bar).item())
    if cut < 0:
        raise MusicXLError(f'the sequence has fewer than {n_bar + 1} bars')     # the reference raises IndexError here
    return torch.cat([ids[:cut], ids.new_tensor([sob_token_id])])
