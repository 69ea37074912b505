"""
ORACLE (test infrastructure, NOT product code) -- CPU fp32 restatement of the Transformer-XL path
of StefanHeng/Symbolic-Music-Generation.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
The product path (`symbolic_music_generation_amd/`) never does.

PARITY STATUS: **decoder layer and adaptive softmax pinned on external implementations, the glue around them unpinned**.
Pinned (tests/test_xlnet_pin_cpu.py; goldens from tests/golden/make_xlnet_relattn_goldens.py, inputs + outputs only):
  * sinusoid [sin || cos] table with clamp_len, the pad/view rel-shift, `RelPartialLearnableMultiHeadAttn`'s AC + BD -> scale ->
    same_length mask -> softmax -> .V  against HuggingFace XLNet's `relative_positional_encoding` / `rel_shift_bnij` /
    `rel_attn_core` (installed transformers 5.15; XLNet inherits Transformer-XL's relative attention unchanged);
  * the whole `DecoderLayer` (qkv_net over cat(mems, h), r_net, o_net, post-LN residual, relu FFN, post-LN) against a whole
    HuggingFace `XLNetLayer` carrying the same weights;
  * `ProjectedAdaptiveLogSoftmax` (div_val = 1) against `torch.nn.AdaptiveLogSoftmaxWithLoss`.
Unpinned: the glue -- embedding scale, dropout placement, mems update, the reference's label guard and loss reduction (these
follow the reference's own file line by line) -- for the reason below.  The arithmetic of this path lives in the un-vendored third-party
dependency `transformers==4.25.1` (`/root/reference/requirements.txt:150`), modules
`transformers/models/transfo_xl/modeling_transfo_xl.py` and `modeling_transfo_xl_utilities.py`.  That
package is absent from /root/reference and from this image (transformers 5.15 dropped the model), and
the reference holds no golden logits/loss for it.  This file restates the published upstream algorithm
(Dai et al. 2019 + the HF 4.25.1 behaviour documented in SURVEY.md Appendix A) in the *reference style*
(dense (qlen, klen) score einsums, materialised pad/view rel-shift, fp32, zero-initialised mems,
`same_length` mask) and is anchored on the reference's own call sites:
  - config presets / derived fields        musicnlp/models/transformer_xl.py:15-77
  - head forward, loss reduction           musicnlp/models/transformer_xl.py:130-221
  - generation input contract              musicnlp/models/transformer_xl.py:223-241
  - known answers: 92 435 362 parameters for (base, V=418)  notebook/train/transformer-xl.ipynb:491
State-dict names follow upstream (`transformer.layers.N.dec_attn.qkv_net.weight`, ...), SURVEY A.7.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

PT_LOSS_PAD = -100  # musicnlp/util/train/train_util_wrap.py:22


# --------------------------------------------------------------------------------------------
# config: musicnlp/models/transformer_xl.py:15-77 on top of upstream TransfoXLConfig defaults
# (defaults visible in the logged config, notebook/train/transformer-xl.ipynb cell 10)
# --------------------------------------------------------------------------------------------
PRESETS = {  # transformer_xl.py:16-23
    'debug': dict(d_model=128, n_head=8, n_layer=4),
    'debug-large': dict(d_model=128, n_head=8, n_layer=4),
    'tiny': dict(d_model=256, n_head=8, n_layer=6),
    'small': dict(d_model=512, n_head=8, n_layer=12),
    'base': dict(d_model=768, n_head=12, n_layer=12),
    'large': dict(d_model=1024, n_head=16, n_layer=18),
}
SIZE2MAX_LENGTH = {'debug': 64, 'debug-large': 128, 'tiny': 512, 'small': 1024, 'base': 2048, 'large': 2048}


def cutoffs_for_vocab(vsz: int) -> List[int]:
    """transformer_xl.py:56-66"""
    if vsz >= 32768 * 8:
        return [20000, 40000, 200000]
    if vsz >= 32768:
        return [10000]
    if vsz >= 16384:
        return [5000]
    if vsz >= 1000:
        return [1000]
    return []


@dataclass
class RefXLConfig:
    vocab_size: int = 1190
    d_model: int = 768
    n_head: int = 12
    n_layer: int = 12
    d_head: int = 64
    d_inner: int = 3072
    d_embed: int = 768
    mem_len: int = 256
    clamp_len: int = 1024
    cutoffs: List[int] = field(default_factory=list)
    div_val: int = 1
    same_length: bool = True
    dropout: float = 0.1
    dropatt: float = 0.0
    layer_norm_epsilon: float = 1e-5
    init_std: float = 0.02
    proj_init_std: float = 0.01
    eos_token_id: int = 0
    max_length_: int = 2048

    @staticmethod
    def from_preset(model_size: str = 'base', vocab_size: Optional[int] = None, max_length: Optional[int] = None,
                    **kwargs) -> 'RefXLConfig':
        p = dict(PRESETS[model_size])
        d, h = p['d_model'], p['n_head']
        assert d % h == 0
        if 'debug' in model_size:
            m_len, c_len = 64, 64  # transformer_xl.py:30-31
        else:
            m_len = max(128, SIZE2MAX_LENGTH[model_size] // 8)  # :33
            c_len = max(1024, SIZE2MAX_LENGTH[model_size] // 2)  # :34
        p.update(d_embed=d, d_inner=4 * d, d_head=d // h, mem_len=m_len, clamp_len=c_len, div_val=1)
        if vocab_size is not None:
            p['vocab_size'] = vocab_size
            p['cutoffs'] = cutoffs_for_vocab(vocab_size)
        p.update(kwargs)  # :67  kwargs override everything
        p['max_length_'] = max_length or SIZE2MAX_LENGTH[model_size]  # :70
        return RefXLConfig(**p)


# --------------------------------------------------------------------------------------------
# upstream modules
# --------------------------------------------------------------------------------------------
class PositionalEmbedding(nn.Module):
    """A.2: inv_freq = 10000^(-2k/d); pos_emb = cat(sin, cos) (halves, not interleaved)."""

    def __init__(self, demb: int):
        super().__init__()
        inv_freq = 1 / (10000 ** (torch.arange(0.0, demb, 2.0) / demb))
        self.register_buffer('inv_freq', inv_freq)

    def forward(self, pos_seq: torch.Tensor) -> torch.Tensor:
        sinusoid_inp = torch.outer(pos_seq, self.inv_freq)
        pos_emb = torch.cat([sinusoid_inp.sin(), sinusoid_inp.cos()], dim=-1)
        return pos_emb[:, None, :]


class PositionwiseFF(nn.Module):
    """A.5: h <- LN(h + drop(W2 drop(relu(W1 h + b1)) + b2))  (post-LN)."""

    def __init__(self, d_model, d_inner, dropout, eps):
        super().__init__()
        self.CoreNet = nn.Sequential(
            nn.Linear(d_model, d_inner), nn.ReLU(inplace=False), nn.Dropout(dropout),
            nn.Linear(d_inner, d_model), nn.Dropout(dropout),
        )
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)

    def forward(self, inp):
        return self.layer_norm(inp + self.CoreNet(inp))


class RelPartialLearnableMultiHeadAttn(nn.Module):
    """A.3; time-major (len, B, d) like upstream."""

    def __init__(self, n_head, d_model, d_head, dropout, dropatt, eps):
        super().__init__()
        self.n_head, self.d_model, self.d_head = n_head, d_model, d_head
        self.qkv_net = nn.Linear(d_model, 3 * n_head * d_head, bias=False)
        self.drop = nn.Dropout(dropout)
        self.dropatt = nn.Dropout(dropatt)
        self.o_net = nn.Linear(n_head * d_head, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=eps)
        self.scale = 1 / (d_head ** 0.5)
        self.r_r_bias = nn.Parameter(torch.zeros(n_head, d_head))  # untie_r=True -> per layer
        self.r_w_bias = nn.Parameter(torch.zeros(n_head, d_head))
        self.r_net = nn.Linear(d_model, n_head * d_head, bias=False)

    @staticmethod
    def _rel_shift(x):
        # pad one zero column on the key axis, view as (klen+1, qlen), drop first row, view back
        zero_pad = torch.zeros((x.size(0), 1) + x.size()[2:], device=x.device, dtype=x.dtype)
        x_padded = torch.cat([zero_pad, x], dim=1)
        x_padded = x_padded.view((x.size(1) + 1, x.size(0)) + x.size()[2:])
        return x_padded[1:].view_as(x)

    def forward(self, w, r, attn_mask, mems):
        qlen, rlen, bsz = w.size(0), r.size(0), w.size(1)
        cat = torch.cat([mems, w], 0)
        w_heads = self.qkv_net(cat)
        r_head_k = self.r_net(r)
        w_head_q, w_head_k, w_head_v = torch.chunk(w_heads, 3, dim=-1)
        w_head_q = w_head_q[-qlen:]
        klen = w_head_k.size(0)
        w_head_q = w_head_q.view(qlen, bsz, self.n_head, self.d_head)
        w_head_k = w_head_k.view(klen, bsz, self.n_head, self.d_head)
        w_head_v = w_head_v.view(klen, bsz, self.n_head, self.d_head)
        r_head_k = r_head_k.view(rlen, self.n_head, self.d_head)

        rw_head_q = w_head_q + self.r_w_bias
        AC = torch.einsum('ibnd,jbnd->ijbn', rw_head_q, w_head_k)
        rr_head_q = w_head_q + self.r_r_bias
        BD = torch.einsum('ibnd,jnd->ijbn', rr_head_q, r_head_k)
        BD = self._rel_shift(BD)

        attn_score = (AC + BD) * self.scale
        mask_value = torch.finfo(attn_score.dtype).min
        if attn_mask is not None and torch.sum(attn_mask).item():
            attn_score = attn_score.float().masked_fill(attn_mask[:, :, :, None] == 1, mask_value).type_as(attn_score)
        attn_prob = F.softmax(attn_score, dim=1)
        attn_prob = self.dropatt(attn_prob)
        attn_vec = torch.einsum('ijbn,jbnd->ibnd', attn_prob, w_head_v)
        attn_vec = attn_vec.contiguous().view(qlen, bsz, self.n_head * self.d_head)
        attn_out = self.drop(self.o_net(attn_vec))
        return self.layer_norm(w + attn_out)


class DecoderLayer(nn.Module):
    def __init__(self, c: RefXLConfig):
        super().__init__()
        self.dec_attn = RelPartialLearnableMultiHeadAttn(c.n_head, c.d_model, c.d_head, c.dropout, c.dropatt,
                                                         c.layer_norm_epsilon)
        self.pos_ff = PositionwiseFF(c.d_model, c.d_inner, c.dropout, c.layer_norm_epsilon)

    def forward(self, dec_inp, r, dec_attn_mask, mems):
        return self.pos_ff(self.dec_attn(dec_inp, r, dec_attn_mask, mems))


class AdaptiveEmbedding(nn.Module):
    """A.1 (div_val=1, d_embed == d_model): Embedding(V, d)[ids] * sqrt(d)."""

    def __init__(self, n_token, d_embed, d_proj):
        super().__init__()
        assert d_embed == d_proj, 'reference always sets d_embed = d_model (transformer_xl.py:36)'
        self.emb_scale = d_proj ** 0.5
        self.emb_layers = nn.ModuleList([nn.Embedding(n_token, d_embed)])

    def forward(self, inp):
        return self.emb_layers[0](inp) * self.emb_scale


class ProjectedAdaptiveLogSoftmax(nn.Module):
    """A.6 (div_val=1).  Returns per-token NLL (with labels; shift inside) or full log-probs."""

    def __init__(self, n_token, d_embed, d_proj, cutoffs):
        super().__init__()
        self.n_token = n_token
        self.cutoffs = list(cutoffs) + [n_token]
        self.cutoff_ends = [0] + self.cutoffs
        self.shortlist_size = self.cutoffs[0]
        self.n_clusters = len(self.cutoffs) - 1
        self.head_size = self.shortlist_size + self.n_clusters
        if self.n_clusters > 0:
            self.cluster_weight = nn.Parameter(torch.zeros(self.n_clusters, d_embed))
            self.cluster_bias = nn.Parameter(torch.zeros(self.n_clusters))
        self.out_layers = nn.ModuleList([nn.Linear(d_embed, n_token)])

    def forward(self, hidden, labels=None, keep_order=False):
        if labels is not None:
            hidden = hidden[..., :-1, :].contiguous()
            labels = labels[..., 1:].contiguous()
            hidden = hidden.view(-1, hidden.size(-1))
            labels = labels.view(-1)
        else:
            hidden = hidden.view(-1, hidden.size(-1))
        W, b = self.out_layers[0].weight, self.out_layers[0].bias
        if self.n_clusters == 0:
            logit = F.linear(hidden, W, b)
            if labels is not None:
                mask = labels != -100
                out = torch.zeros_like(labels, dtype=hidden.dtype)
                out[mask] = -F.log_softmax(logit, dim=-1)[mask].gather(1, labels[mask].unsqueeze(1)).squeeze(1)
            else:
                out = F.log_softmax(logit, dim=-1)
            return out
        weights, biases = [], []
        for i in range(len(self.cutoffs)):
            l_idx, r_idx = self.cutoff_ends[i], self.cutoff_ends[i + 1]
            w_i, b_i = W[l_idx:r_idx], b[l_idx:r_idx]
            if i == 0:
                w_i = torch.cat([w_i, self.cluster_weight], dim=0)
                b_i = torch.cat([b_i, self.cluster_bias], dim=0)
            weights.append(w_i)
            biases.append(b_i)
        head_logprob = F.log_softmax(F.linear(hidden, weights[0], biases[0]), dim=1)
        if labels is None:
            out = hidden.new_empty((head_logprob.size(0), self.n_token))
        else:
            out = torch.zeros_like(labels, dtype=hidden.dtype)
        offset = 0
        cutoff_values = [0] + self.cutoffs
        for i in range(len(cutoff_values) - 1):
            l_idx, r_idx = cutoff_values[i], cutoff_values[i + 1]
            if labels is not None:
                mask_i = (labels >= l_idx) & (labels < r_idx)
                indices_i = mask_i.nonzero().squeeze(1)
                if indices_i.numel() == 0:
                    continue
                target_i = labels.index_select(0, indices_i) - l_idx
                head_logprob_i = head_logprob.index_select(0, indices_i)
                hidden_i = hidden.index_select(0, indices_i)
            else:
                hidden_i = hidden
            if i == 0:
                if labels is not None:
                    logprob_i = head_logprob_i.gather(1, target_i[:, None]).squeeze(1)
                else:
                    out[:, :self.cutoffs[0]] = head_logprob[:, :self.cutoffs[0]]
            else:
                tail_logprob_i = F.log_softmax(F.linear(hidden_i, weights[i], biases[i]), dim=1)
                cluster_prob_idx = self.cutoffs[0] + i - 1
                if labels is not None:
                    logprob_i = head_logprob_i[:, cluster_prob_idx] + tail_logprob_i.gather(1, target_i[:, None]).squeeze(1)
                else:
                    out[:, l_idx:r_idx] = head_logprob[:, cluster_prob_idx, None] + tail_logprob_i
            if labels is not None:
                if keep_order:
                    out.index_copy_(0, indices_i, -logprob_i)
                else:  # upstream default: cluster order (the reference only reduces with `losses != 0`)
                    out[offset:offset + logprob_i.size(0)].copy_(-logprob_i)
                offset += logprob_i.size(0)
        return out


class RefTransfoXLModel(nn.Module):
    def __init__(self, c: RefXLConfig):
        super().__init__()
        self.c = c
        self.word_emb = AdaptiveEmbedding(c.vocab_size, c.d_embed, c.d_model)
        self.drop = nn.Dropout(c.dropout)
        self.layers = nn.ModuleList([DecoderLayer(c) for _ in range(c.n_layer)])
        self.pos_emb = PositionalEmbedding(c.d_model)
        # test-infrastructure switch (no upstream counterpart): recompute each layer in the backward instead of keeping its dense
        # (qlen, klen) score tensors -- same arithmetic, a few GB instead of ~25 GB for the 12-layer / 2048-token autograd
        self.checkpoint_layers = False

    def init_mems(self, bsz):
        p = next(self.parameters())
        return [torch.zeros(self.c.mem_len, bsz, self.c.d_model, dtype=p.dtype, device=p.device)
                for _ in range(self.c.n_layer)]

    def _update_mems(self, hids, mems, mlen, qlen):
        with torch.no_grad():
            new_mems = []
            end_idx = mlen + max(0, qlen)
            beg_idx = max(0, end_idx - self.c.mem_len)
            for i in range(len(hids)):
                cat = torch.cat([mems[i], hids[i]], dim=0)
                new_mems.append(cat[beg_idx:end_idx].detach())
        return new_mems

    def forward(self, input_ids, mems=None):
        input_ids = input_ids.transpose(0, 1).contiguous()  # API (B, T) -> internal (T, B)
        qlen, bsz = input_ids.size()
        if mems is None:
            mems = self.init_mems(bsz)
        word_emb = self.word_emb(input_ids)
        mlen = mems[0].size(0)
        klen = mlen + qlen
        assert self.c.same_length
        all_ones = word_emb.new_ones((qlen, klen), dtype=torch.uint8)
        mask_len = klen - self.c.mem_len
        mask_shift_len = qlen - mask_len if mask_len > 0 else qlen
        dec_attn_mask = (torch.triu(all_ones, 1 + mlen) + torch.tril(all_ones, -mask_shift_len))[:, :, None]
        pos_seq = torch.arange(klen - 1, -1, -1.0, dtype=word_emb.dtype)
        if self.c.clamp_len > 0:
            pos_seq.clamp_(max=self.c.clamp_len)
        pos_emb = self.drop(self.pos_emb(pos_seq))
        core_out = self.drop(word_emb)
        hids = []
        for i, layer in enumerate(self.layers):
            hids.append(core_out)
            if self.checkpoint_layers and torch.is_grad_enabled():
                from torch.utils.checkpoint import checkpoint
                core_out = checkpoint(layer, core_out, pos_emb, dec_attn_mask, mems[i], use_reentrant=False)
            else:
                core_out = layer(core_out, pos_emb, dec_attn_mask, mems[i])
        core_out = self.drop(core_out)
        new_mems = self._update_mems(hids, mems, mlen, qlen)
        return core_out.transpose(0, 1).contiguous(), new_mems


@dataclass
class RefXLOutput:
    loss: Optional[torch.Tensor] = None
    prediction_scores: Optional[torch.Tensor] = None
    losses: Optional[torch.Tensor] = None
    mems: Optional[List[torch.Tensor]] = None

    @property
    def logits(self):  # transformer_xl.py:117-124 -- log-probabilities
        return self.prediction_scores


class RefTransfoXLLMHeadModel(nn.Module):
    """musicnlp/models/transformer_xl.py:127-241 restated on top of the upstream restatement above."""

    def __init__(self, c: RefXLConfig):
        super().__init__()
        self.config = c
        self.transformer = RefTransfoXLModel(c)
        self.crit = ProjectedAdaptiveLogSoftmax(c.vocab_size, c.d_embed, c.d_model, c.cutoffs)
        self.apply(self._init_weights)
        # tie_word_embeddings=True: weight tied to the embedding, own bias (A.6)
        self.crit.out_layers[0].weight = self.transformer.word_emb.emb_layers[0].weight

    def _init_weights(self, m):
        std = self.config.init_std
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0.0, std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.0)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, 0.0, std)
        elif isinstance(m, nn.LayerNorm):
            nn.init.normal_(m.weight, 1.0, std)
            nn.init.constant_(m.bias, 0.0)
        elif isinstance(m, ProjectedAdaptiveLogSoftmax):
            if m.n_clusters > 0:
                nn.init.normal_(m.cluster_weight, 0.0, std)
                nn.init.constant_(m.cluster_bias, 0.0)
        elif isinstance(m, RelPartialLearnableMultiHeadAttn):
            nn.init.normal_(m.r_w_bias, 0.0, std)
            nn.init.normal_(m.r_r_bias, 0.0, std)

    def forward(self, input_ids, mems=None, labels=None) -> RefXLOutput:
        bsz, tgt_len = input_ids.size(0), input_ids.size(1)
        last_hidden, new_mems = self.transformer(input_ids, mems=mems)
        pred_hid = last_hidden[:, -tgt_len:]
        if labels is not None:
            labels = labels.clone()
            # transformer_xl.py:176-182: all-ignored guard on row 0
            if labels[0, 1:].sum() == (labels.size(1) - 1) * -100:
                labels[0, 1] = self.config.eos_token_id
        softmax_output = self.crit(pred_hid, labels)  # :185
        _softmax_output = softmax_output
        in_eval = not self.training
        if in_eval and labels is not None:
            _softmax_output = self.crit(pred_hid, None)  # :193
        prediction_scores = _softmax_output.view(bsz, tgt_len, -1) if (labels is None or in_eval) else None
        if labels is not None:
            losses = softmax_output.view(bsz, tgt_len - 1)
            loss = losses[losses != 0].mean()  # :200
        else:
            losses, loss = None, None
        return RefXLOutput(loss=loss, prediction_scores=prediction_scores, losses=losses, mems=new_mems)

    @staticmethod
    def prepare_inputs_for_generation(input_ids, past=None):
        """transformer_xl.py:223-241"""
        if past:
            return dict(mems=past, input_ids=input_ids[:, -1].unsqueeze(-1))
        return dict(input_ids=input_ids)

    @torch.no_grad()
    def greedy_generate(self, input_ids, max_length) -> torch.Tensor:
        """HF 4.25.1 GenerationMixin.greedy_search restated for this model: argmax of the last
        position's log-probs; mems carried through `past` (musicnlp/trainer/eval.py:333 call site)."""
        self.eval()
        past = None
        ids = input_ids
        while ids.size(1) < max_length:
            inp = self.prepare_inputs_for_generation(ids, past)
            out = self(inp['input_ids'], mems=inp.get('mems'))
            nxt = out.prediction_scores[:, -1, :].argmax(-1, keepdim=True)
            past = out.mems
            ids = torch.cat([ids, nxt], dim=1)
        return ids


class _RefBeamHyps:
    """HF 4.25.1 generation/beam_search.py `BeamHypotheses`: the n best finished hypotheses of one batch item, scored
    sum_logprobs / len ** length_penalty"""

    def __init__(self, num_beams, length_penalty, early_stopping):
        self.n, self.lp, self.early = num_beams, length_penalty, early_stopping
        self.beams, self.worst = [], 1e9

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (hyp.shape[-1] ** self.lp)
        if len(self.beams) < self.n or score > self.worst:
            self.beams.append((score, hyp))
            if len(self.beams) > self.n:
                order = sorted((sc, i) for i, (sc, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self.beams) < self.n:
            return False
        if self.early:
            return True
        return self.worst >= best_sum_logprobs / cur_len ** self.lp


@torch.no_grad()
def ref_beam_search(model: 'RefTransfoXLLMHeadModel', input_ids, max_length, num_beams=3, early_stopping=True,
                    length_penalty=1.0, num_return_sequences=1, return_scores=False):
    """HF 4.25.1 `GenerationMixin.beam_search` + `BeamSearchScorer.process / finalize` restated for this model (the
    `strategy='beam'` branch of musicnlp/trainer/eval.py:302-321 with do_sample=False; test infrastructure).  The prompt is
    expanded num_beams times (repeat_interleave), beam 0 of each item starts at score 0 and the others at -1e9, each step takes
    the 2 * num_beams best (beam, token) continuations per item in score order, finished ones (eos) go to the hypothesis heap,
    the first num_beams open ones continue, and the mems follow their beams (`_reorder_cache`: index_select(1, beam_idx)).
    Returns (B * num_return_sequences, L) ids padded with pad = eos as HF does when the config has no pad token."""
    model.eval()
    eos = model.config.eos_token_id
    pad = eos
    B, nb = input_ids.shape[0], num_beams
    ids = input_ids.repeat_interleave(nb, 0)
    beam_scores = torch.zeros(B, nb)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_RefBeamHyps(nb, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    past = None
    while True:
        inp = model.prepare_inputs_for_generation(ids, past)
        out = model(inp['input_ids'], mems=inp.get('mems'))
        logp = out.prediction_scores[:, -1, :]                   # already log-probabilities
        V = logp.shape[-1]
        sc = (logp + beam_scores[:, None]).view(B, nb * V)
        top_s, top_i = sc.topk(2 * nb, dim=1, largest=True, sorted=True)
        top_b, top_t = top_i // V, top_i % V
        cur_len = ids.shape[-1]
        n_scores, n_tok, n_idx = torch.zeros(B, nb), torch.zeros(B, nb, dtype=torch.long), torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                n_tok[b] = pad
                continue
            k = 0
            for rank in range(2 * nb):
                tok, s_, src = int(top_t[b, rank]), float(top_s[b, rank]), b * nb + int(top_b[b, rank])
                if tok == eos:
                    if rank >= nb:
                        continue
                    hyps[b].add(ids[src].clone(), s_)
                else:
                    n_scores[b, k], n_tok[b, k], n_idx[b, k] = s_, tok, src
                    k += 1
                if k == nb:
                    break
            assert k == nb
            done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
        beam_scores, beam_tok, beam_idx = n_scores.view(-1), n_tok.view(-1), n_idx.view(-1)
        ids = torch.cat([ids[beam_idx], beam_tok[:, None]], 1)
        past = [m.index_select(1, beam_idx) for m in out.mems]
        if all(done) or ids.shape[-1] >= max_length:
            break
    for b in range(B):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(ids[b * nb + j], float(beam_scores[b * nb + j]))
    best, scores = [], []
    for b in range(B):
        srt = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(num_return_sequences):
            sc_, h = srt.pop()
            best.append(h); scores.append(sc_)
    lens = torch.tensor([len(h) for h in best])
    L = min(int(lens.max()) + 1, max_length)
    outp = torch.full((len(best), L), pad, dtype=torch.long)
    for i, h in enumerate(best):
        outp[i, :len(h)] = h
        if len(h) < L:
            outp[i, len(h)] = eos
    return (outp, torch.tensor(scores)) if return_scores else outp


@torch.no_grad()
def ref_group_beam_search(model: 'RefTransfoXLLMHeadModel', input_ids, max_length, num_beams=4, num_beam_groups=2,
                          diversity_penalty=0.0, early_stopping=True, length_penalty=1.0, num_return_sequences=1,
                          return_scores=False):
    """HF 4.25.1 `GenerationMixin.group_beam_search` + `BeamSearchScorer(num_beam_groups=)` + `HammingDiversityLogitsProcessor`
    restated for this model (musicnlp/trainer/eval.py:303-317; test infrastructure).  Every step runs the model on ALL beams with
    carried mems; group g's log-probabilities lose diversity_penalty x the number of beams of the earlier groups of the same
    item that have just emitted a token (torch.bincount of `current_tokens`), get the group's running beam scores added, and the
    2 * group_size best continuations per item go through the scorer (4.25.1: ONE BeamHypotheses of capacity num_beams and one
    done flag per item, shared by the groups).  `reordering_indices` maps every row to the row whose mems it inherits."""
    model.eval()
    eos = model.config.eos_token_id
    pad = eos
    B, nb, ng = input_ids.shape[0], num_beams, num_beam_groups
    assert nb % ng == 0
    gs = nb // ng
    ids = input_ids.repeat_interleave(nb, 0)
    beam_scores = torch.full((B, nb), -1e9)
    beam_scores[:, ::gs] = 0
    beam_scores = beam_scores.view(-1)
    hyps = [_RefBeamHyps(nb, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    past = None
    while True:
        inp = model.prepare_inputs_for_generation(ids, past)
        out = model(inp['input_ids'], mems=inp.get('mems'))
        logp_all = out.prediction_scores[:, -1, :]
        V = logp_all.shape[-1]
        cur_len = ids.shape[-1]
        current = torch.zeros(B * nb, dtype=torch.long)
        reorder = torch.zeros(B * nb, dtype=torch.long)
        new_ids = ids.clone()
        for g in range(ng):
            g0, g1 = g * gs, (g + 1) * gs
            rows = [b * nb + j for b in range(B) for j in range(g0, g1)]
            sc = logp_all[rows].clone()
            if diversity_penalty > 0 and g > 0:
                for b in range(B):
                    prev = current[b * nb:b * nb + g0]
                    freq = torch.bincount(prev, minlength=V).to(sc.dtype)
                    sc[b * gs:(b + 1) * gs] -= diversity_penalty * freq
            sc = (sc + beam_scores[rows][:, None]).view(B, gs * V)
            top_s, top_i = sc.topk(2 * gs, dim=1, largest=True, sorted=True)
            top_b, top_t = top_i // V, top_i % V
            for b in range(B):
                if done[b]:
                    for j in range(gs):
                        r = b * nb + g0 + j
                        beam_scores[r], current[r], reorder[r] = 0.0, pad, r
                    continue
                k = 0
                for rank in range(2 * gs):
                    tok, s_, src = int(top_t[b, rank]), float(top_s[b, rank]), b * nb + g0 + int(top_b[b, rank])
                    if tok == eos:
                        if rank >= gs:
                            continue
                        hyps[b].add(ids[src].clone(), s_)
                    else:
                        r = b * nb + g0 + k
                        beam_scores[r], current[r], reorder[r] = s_, tok, src
                        k += 1
                    if k == gs:
                        break
                assert k == gs
                done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len)
        ids = torch.cat([ids[reorder], current[:, None]], 1)
        past = [m.index_select(1, reorder) for m in out.mems]
        if all(done) or ids.shape[-1] >= max_length:
            break
    for b in range(B):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(ids[b * nb + j], float(beam_scores[b * nb + j]))
    best, scores = [], []
    for b in range(B):
        srt = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(num_return_sequences):
            sc_, h = srt.pop()
            best.append(h); scores.append(sc_)
    L = min(max(len(h) for h in best) + 1, max_length)
    outp = torch.full((len(best), L), pad, dtype=torch.long)
    for i, h in enumerate(best):
        outp[i, :len(h)] = h
        if len(h) < L:
            outp[i, len(h)] = eos
    return (outp, torch.tensor(scores)) if return_scores else outp


@torch.no_grad()
def ref_contrastive_search(model: 'RefTransfoXLLMHeadModel', input_ids, max_length, top_k=4, penalty_alpha=0.6, return_trace=False):
    """HF 4.25.1 `GenerationMixin.contrastive_search` + `_ranking_fast` restated for this model (the reference's 'contrastive'
    strategy, musicnlp/trainer/eval.py:296-302, with `past` = the mems, musicnlp/models/transformer_xl.py:223-241; test
    infrastructure).  Prompt pass -> last-layer hidden states of every position + next-token log-probs; then per step: top-k
    filter (TopKLogitsWarper) and softmax -> k candidates with their probabilities; one forward of the B*k candidates on the
    k-times replicated mems; score = (1 - alpha) * p - alpha * max cosine(candidate hidden, every context hidden); the winner's
    token, hidden state, mems and log-probs carry on.  (eos = 0 = [OMIT] practically never fires; it ends a row as in HF.)"""
    model.eval()
    eos = model.config.eos_token_id
    B, K = input_ids.shape[0], top_k
    hid_full, mems = model.transformer(input_ids, mems=None)                 # (B, T, d): hidden_states[-1] in eval mode
    logp = model.crit(hid_full[:, -1:], None).view(B, -1)                    # log-probs of the next token
    ctx = hid_full
    ids = input_ids.clone()
    unfinished = torch.ones(B, dtype=torch.bool)
    trace = []
    while ids.shape[1] < max_length:
        filt = logp.masked_fill(logp < logp.topk(K, -1).values[:, -1:], float('-inf'))
        probs = filt.softmax(-1)
        top_p, top_i = probs.topk(K, -1)
        cand = top_i.reshape(-1, 1)
        mem_k = [m.repeat_interleave(K, dim=1) for m in mems]                # mems are (M, B, d): batch on dim 1
        h_c, mems_c = model.transformer(cand, mems=mem_k)                    # (B*K, 1, d)
        logp_c = model.crit(h_c, None).view(B * K, -1)
        ctx_k = ctx.repeat_interleave(K, 0)
        a = ctx_k / ctx_k.norm(dim=2, keepdim=True)
        b_ = h_c / h_c.norm(dim=2, keepdim=True)
        pen = torch.matmul(a, b_.transpose(1, 2)).squeeze(-1).max(-1).values
        score = ((1.0 - penalty_alpha) * top_p.reshape(-1) - penalty_alpha * pen).view(B, K)
        sel = score.argmax(-1)
        trace.append(dict(score=score.clone(), sel=sel.clone(), cand=top_i.clone()))
        rows = torch.arange(B) * K + sel
        tok = top_i[torch.arange(B), sel]
        tok = torch.where(unfinished, tok, torch.full_like(tok, eos))
        ids = torch.cat([ids, tok[:, None]], 1)
        unfinished = unfinished & (tok != eos)
        ctx = torch.cat([ctx, h_c[rows]], 1)
        mems = [m.index_select(1, rows) for m in mems_c]
        logp = logp_c[rows]
        if not unfinished.any():
            break
    return (ids, trace) if return_trace else ids


def count_parameters(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())  # shared (tied) tensors counted once
