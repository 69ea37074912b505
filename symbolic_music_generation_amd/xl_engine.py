"""Transformer-XL execution engine: one flat parameter buffer, an explicit layer loop that enqueues the HIP kernels of
libmusicxl (no autograd tape, no tracing), saved activations in a pre-allocated workspace.

Semantics follow upstream `TransfoXLModel` / `TransfoXLLMHeadModel` of transformers==4.25.1 as wrapped by
musicnlp/models/transformer_xl.py:127-241: post-LN layers, same_length window of `mem_len` keys,
zero initial mems, tied embedding/softmax weight with own bias, adaptive softmax with optional cutoffs,
loss = mean over non-zero per-token NLLs.  Internal layout is batch-major (B, T, d) instead of upstream's (T, B, d).

Precision: bf16 storage / MFMA operands with fp32 accumulation; fp32 master weights, gradients and Adam moments.
"""
import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import MusicXLError

F = ops  # flags live there


def _r8(n: int) -> int:
    return (n + 7) // 8 * 8


class ParamLayout:
    """Flat layout: [decay segment | no-decay segment]; every tensor starts at a multiple of 8 elements.
    HF Trainer decays everything except names containing 'bias' and LayerNorm weights (trainer.py create_optimizer)."""

    def __init__(self, cfg):
        d, Fi, H, dh, V, L = cfg.d_model, cfg.d_inner, cfg.n_head, cfg.d_head, cfg.vocab_size, cfg.n_layer
        ncl = len(cfg.cutoffs)
        self.n_head_rows = V + ncl
        self.head_rows_padded = _r8(V + ncl)
        decay: List[Tuple[str, Tuple[int, ...]]] = [('transformer.word_emb.emb_layers.0.weight', (V, d))]
        if ncl:
            decay.append(('crit.cluster_weight', (ncl, d)))
        if self.head_rows_padded > V + ncl:
            decay.append(('_pad.head_rows', (self.head_rows_padded - V - ncl, d)))
        nodecay: List[Tuple[str, Tuple[int, ...]]] = [('crit.out_layers.0.bias', (V,))]
        if ncl:
            nodecay.append(('crit.cluster_bias', (ncl,)))
        for l in range(L):
            p = f'transformer.layers.{l}.'
            decay += [(p + 'dec_attn.qkv_net.weight', (3 * H * dh, d)), (p + 'dec_attn.r_net.weight', (H * dh, d)),
                      (p + 'dec_attn.o_net.weight', (d, H * dh)), (p + 'pos_ff.CoreNet.0.weight', (Fi, d)),
                      (p + 'pos_ff.CoreNet.3.weight', (d, Fi))]
            nodecay += [(p + 'dec_attn.r_r_bias', (H, dh)), (p + 'dec_attn.r_w_bias', (H, dh)),
                        (p + 'dec_attn.layer_norm.weight', (d,)), (p + 'dec_attn.layer_norm.bias', (d,)),
                        (p + 'pos_ff.CoreNet.0.bias', (Fi,)), (p + 'pos_ff.CoreNet.3.bias', (d,)),
                        (p + 'pos_ff.layer_norm.weight', (d,)), (p + 'pos_ff.layer_norm.bias', (d,))]
        self.entries: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        off = 0
        for name, shape in decay:
            self.entries[name] = (off, shape)
            n = math.prod(shape)
            # the head rows [E ; cluster_weight ; pad] must stay contiguous: no padding between them
            contiguous_next = name in ('transformer.word_emb.emb_layers.0.weight', 'crit.cluster_weight')
            off = off + n if contiguous_next else _r8(off + n)
        off = _r8(off)
        self.n_decay = off
        for name, shape in nodecay:
            self.entries[name] = (off, shape)
            n = math.prod(shape)
            off = off + n if (name == 'crit.out_layers.0.bias' and ncl) else _r8(off + n)
        self.total = _r8(off)

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.entries[name]
        return buf[off:off + math.prod(shape)].view(*shape)

    def real_names(self):
        return [n for n in self.entries if not n.startswith('_pad.')]


class _WS:
    pass


class XLEngine:
    SITE_EMB, SITE_POS, SITE_FINAL = 0, 1, 2

    def __init__(self, cfg, device, seed: int = 77):
        if not torch.cuda.is_available():
            raise MusicXLError('XLEngine needs a GPU: the product path has no CPU fallback')
        self.cfg = cfg
        self.dev = torch.device(device)
        assert cfg.d_model % 8 == 0 and cfg.d_inner % 8 == 0 and cfg.n_head * cfg.d_head == cfg.d_model
        assert cfg.d_head in (16, 32, 64), 'HIP attention kernels cover d_head 16/32/64 (all reference presets)'
        assert cfg.mem_len % 8 == 0 and cfg.mem_len > 0
        assert cfg.same_length and cfg.div_val == 1 and cfg.d_embed == cfg.d_model
        self.layout = ParamLayout(cfg)
        n = self.layout.total
        self.P = torch.zeros(n, device=self.dev, dtype=torch.float32)      # master weights
        self.W = torch.zeros(n, device=self.dev, dtype=torch.bfloat16)     # MFMA operands
        self.G: Optional[torch.Tensor] = None                              # grads (allocated on first train step)
        self.m: Optional[torch.Tensor] = None
        self.v: Optional[torch.Tensor] = None
        self.WT: Optional[Dict[str, torch.Tensor]] = None                  # [in][out] weight copies (training only)
        self.step_count = 0        # optimizer steps taken (AdamW bias correction)
        self.rng_step = 0          # dropout-mask / LSH-rotation stream position: advances with every backward pass
        self.base_seed = seed
        self._ws: Dict[Tuple, _WS] = {}
        self._sumsq = torch.zeros(1, device=self.dev)
        # large vocabularies (the reference's cutoff policy starts clustering for real at V >= 16384, transformer_xl.py:53-66):
        # the head runs cluster by cluster over bucketed tokens (csrc/head_large.hip) instead of one (tokens, V + clusters) GEMM
        self.bucketed_head = len(cfg.cutoffs) > 0 and cfg.vocab_size >= 16384
        self._hb = None            # bucketed-head scratch (per token count)
        self.init_weights(seed)

    # ------------------------------------------------------------------ parameters
    def p32(self, name):
        return self.layout.view(self.P, name)

    def w16(self, name):
        return self.layout.view(self.W, name)

    def g32(self, name):
        return self.layout.view(self.G, name)

    def _lw(self, l, suffix, buf=None):
        return self.layout.view(self.W if buf is None else buf, f'transformer.layers.{l}.{suffix}')

    def init_weights(self, seed: int):
        """upstream `_init_weights`: N(0, init_std) linears / embedding / r_*_bias / cluster_weight, LayerNorm weight
        N(1, init_std), biases 0."""
        g = torch.Generator().manual_seed(seed)
        std = self.cfg.init_std
        host = torch.zeros(self.layout.total)
        for name in self.layout.real_names():
            off, shape = self.layout.entries[name]
            n = math.prod(shape)
            if name.endswith('layer_norm.weight'):
                host[off:off + n] = 1.0 + std * torch.randn(n, generator=g)
            elif name.endswith('.bias') and 'r_r_bias' not in name and 'r_w_bias' not in name:
                pass
            elif name.endswith('cluster_bias'):
                pass
            else:
                host[off:off + n] = std * torch.randn(n, generator=g)
        self.P.copy_(host)
        self.sync_weights()

    def sync_weights(self):
        ops.cast_bf16(self.P, self.W)
        self._refresh_wt()

    # [in][out] copies of the layer Linear weights: dX = dY W then runs in the K-contiguous GEMM form (the DMA-fed large-tile
    # kernel) instead of reading W through transposed LDS fragments.  (L, in, out) per weight kind, refreshed by four batched
    # transposes whenever W changes; 2 x 14 MB per layer at C3.
    _WT_KINDS = ('dec_attn.qkv_net.weight', 'dec_attn.o_net.weight', 'pos_ff.CoreNet.0.weight', 'pos_ff.CoreNet.3.weight')

    def _refresh_wt(self, allocate: bool = False):
        if self.WT is None and not allocate:
            return
        L = self.cfg.n_layer
        if self.WT is None:
            self.WT = {}
            for kind in self._WT_KINDS:
                o, i = self.layout.entries[f'transformer.layers.0.{kind}'][1]
                self.WT[kind] = torch.empty(L, i, o, device=self.dev, dtype=torch.bfloat16)
        for kind in self._WT_KINDS:
            off0, (o, i) = self.layout.entries[f'transformer.layers.0.{kind}']
            stride = (self.layout.entries[f'transformer.layers.1.{kind}'][0] - off0) if L > 1 else 0
            assert all(self.layout.entries[f'transformer.layers.{l}.{kind}'][0] == off0 + l * stride for l in range(L))
            ops.transpose(self.W[off0:], self.WT[kind], o, i, batch=L, src_bstride=stride, dst_bstride=i * o)
        if not self.bucketed_head:       # the head's rows as [in][out], zero beyond the last row (K of the input-gradient product)
            nrow_p, d = self.layout.head_rows_padded, self.cfg.d_model
            if 'head' not in self.WT:
                self.WT['head'] = torch.zeros(d, self._head_kp(), device=self.dev, dtype=torch.bfloat16)
            ops.transpose(self.W[:nrow_p * d].view(nrow_p, d), self.WT['head'], nrow_p, d, ld_dst=self._head_kp())

    def _lwt(self, l, kind):
        return self.WT[kind][l]

    def _head_kp(self):
        return (self.layout.head_rows_padded + 63) // 64 * 64

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict()
        for name in self.layout.real_names():
            sd[name] = self.p32(name).detach().cpu().clone()
        sd['crit.out_layers.0.weight'] = sd['transformer.word_emb.emb_layers.0.weight']  # tied
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = []
        for name in self.layout.real_names():
            if name in sd:
                self.p32(name).copy_(sd[name].to(torch.float32).reshape(self.layout.entries[name][1]))
            else:
                missing.append(name)
        if strict and missing:
            raise KeyError(f'missing parameters: {missing[:4]}...')
        self.sync_weights()

    def num_parameters(self) -> int:
        return sum(math.prod(self.layout.entries[n][1]) for n in self.layout.real_names())

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, T, Kc, train) -> _WS:
        key = (B, T, Kc, train)
        ws = self._ws.get(key)
        if ws is not None:
            self._ws[key] = self._ws.pop(key)      # most recently used last
            return ws
        # bounded cache: a ragged last batch or a generation loop with a growing T must not pile up workspaces (14.5 GB per
        # key at 12L/768d, B = 32); keep the two most recent shapes per mode
        same = [k for k in self._ws if k[-1] == train]
        for k in same[:max(0, len(same) - 1)]:
            del self._ws[k]
        c, dev = self.cfg, self.dev
        d, Fi, H, L, M = c.d_model, c.d_inner, c.n_head, c.n_layer, c.mem_len
        N = B * T
        bf = dict(device=dev, dtype=torch.bfloat16)
        f32 = dict(device=dev, dtype=torch.float32)
        ws = _WS()
        keep = L if train else 1   # eval reuses one slot per activation kind
        ws.phi = torch.empty(M, d, **bf)
        ws.h = [torch.empty(N, d, **bf) for _ in range(L + 1)] if train else [torch.empty(N, d, **bf) for _ in range(2)]
        ws.cat = [torch.empty(B, Kc, d, **bf) for _ in range(keep)] if Kc > T else None
        ws.qkv = [torch.empty(B, Kc, 3 * d, **bf) for _ in range(keep)]
        ws.rd = [torch.empty(M, d, **bf) for _ in range(keep)]
        ws.av = [torch.empty(N, d, **bf) for _ in range(keep)]
        ws.lse = [torch.empty(B, H, T, **f32) for _ in range(keep)]
        # zero-memory training: the forward's phantom value-sum per layer (ops.relattn_fwd(..., oph=, mph=)), which spares the
        # query-owner backward its walk over the all-phantom distance blocks
        # the fused attention backward (ops.relattn_bwd_fused) at the shapes it takes; it wants the value-sum over every phantom cell
        ws.fused_bwd = train and ops.fused_bwd_applies(T=T, dh=c.d_head, M=M, Kc=Kc, B=B, H=H)
        use_oph = train and ((ws.fused_bwd and Kc < M + T) or ops.phantom_sum_applies(T=T, dh=c.d_head, M=M, Kc=Kc))
        ws.oph = [torch.empty(N, d, **bf) for _ in range(keep)] if use_oph else None
        ws.mph = [torch.empty(B, H, T, **f32) for _ in range(keep)] if use_oph else None
        # ... and, with the fused backward, the per-tile records its phantom-cell dRd kernel reads (q + r_r_bias rows, -lse2)
        ws.ph = ([torch.empty(int(ops.lib().mxl_relattn_drd_phantom_ws_bytes(B, T, H)), device=self.dev, dtype=torch.uint8)
                  for _ in range(keep)] if (use_oph and ws.fused_bwd) else None)
        ws.tmp = torch.empty(N, d, **bf)
        ws.h1 = [torch.empty(N, d, **bf) for _ in range(keep)]
        ws.a = [torch.empty(N, Fi, **bf) for _ in range(keep)]
        # relu(+dropout) mask of the FFN activations as bits in the GEMM's own tile layout (one 16-byte load per lane and tile in the
        # backward GEMM's epilogue instead of the bf16 activations row block by row block); None at sizes the large-tile kernel
        # does not take
        mb = ops.gemm_relu_mask_bytes(N, Fi) if train else 0
        ws.rmask = [torch.empty(mb, device=self.dev, dtype=torch.uint8) for _ in range(keep)] if mb else None
        if train:
            ws.z1 = [torch.empty(N, d, **bf) for _ in range(L)]
            ws.z2 = [torch.empty(N, d, **bf) for _ in range(L)]
            ws.st1 = [torch.empty(2, N, **f32) for _ in range(L)]
            ws.st2 = [torch.empty(2, N, **f32) for _ in range(L)]
            ws.hid_d = torch.empty(N, d, **bf)
            # backward scratch
            ws.dA = torch.empty(N, d, **bf)      # gradient stream a (residual path)
            ws.dB = torch.empty(N, d, **bf)      # gradient stream b (GEMM path)
            ws.dC = torch.empty(N, d, **bf)
            ws.dD = torch.empty(N, d, **bf)
            ws.dF = torch.empty(N, Fi, **bf)
            ws.dqkv = torch.empty(B, Kc, 3 * d, **bf)
            ws.delta = torch.empty(B, H, T, **f32)
            if ws.fused_bwd:      # no dG tensor: the partial-dq slabs instead ((M/256 + 1) x (B, T, d) fp32)
                ws.dg = None
                ws.dq_slabs = torch.empty(ops.relattn_bwd_fused_ws_numel(B, T, H, c.d_head, M), **f32)
                ws.qr = None
            else:
                ws.dg = torch.empty(B, H, T, M, **bf)
                ws.qr = torch.empty(B, T, d, **bf)
            ws.d_rd = torch.empty(M, d, **f32)
            ws.d_rd16 = torch.empty(M, d, **bf)
            ws.phi_c = torch.empty(M, d, **bf)
            if not self.bucketed_head:
                # row stride rounded up to the K granule of the large-tile GEMM (the pad columns are written as zeros by
                # mxl_adaptive_nll_bwd*): the input gradient dlogits . W is then an NT product against the [in][out] head copy
                ws.dlogits = torch.empty(N, self._head_kp(), **bf)
                ws.dlogits_lo = torch.empty(N, self._head_kp(), **bf)
        ws.logits = torch.empty(N, self.layout.head_rows_padded, **f32) if not self.bucketed_head else None
        ws.nll = torch.empty(B, max(T - 1, 1), **f32)
        ws.hlse = torch.empty(N, 2, **f32)
        ws.acc = torch.zeros(2, **f32)
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ forward
    def _site(self, l, k):
        return 8 + 8 * l + k

    def forward(self, input_ids: torch.Tensor, mems: Optional[Sequence[torch.Tensor]] = None,
                labels: Optional[torch.Tensor] = None, train: bool = False, want_logprobs: bool = True, kv_sink=None):
        """input_ids (B, T) int64 on device; mems: list of L tensors (B, M, d) bf16 (batch-major) or None (= zero mems).
        Returns dict(loss, losses, logprobs, mems).  In train mode activations are kept for `backward()`."""
        c = self.cfg
        B, T = input_ids.shape
        d, H, dh, L, M, Fi = c.d_model, c.n_head, c.d_head, c.n_layer, c.mem_len, c.d_inner
        V, cut = c.vocab_size, tuple(c.cutoffs)
        N = B * T
        has_mem = mems is not None
        Kc = T + (M if has_mem else 0)
        ws = self._workspace(B, T, Kc, train)
        p = float(c.dropout) if train else 0.0
        seed = ops.mix_seed(self.base_seed, self.rng_step)
        ids = input_ids.contiguous()
        ws.ids, ws.B, ws.T, ws.Kc, ws.p, ws.seed, ws.has_mem = ids, B, T, Kc, p, seed, has_mem

        ops.sinusoid_table(M, d, c.clamp_len, self.dev, drop_p=p, seed=seed, site=self.SITE_POS, out=ws.phi)
        E = self.w16('transformer.word_emb.emb_layers.0.weight')
        ops.embed_fwd(ids, E, ws.h[0], math.sqrt(d), drop_p=p, seed=seed, site=self.SITE_EMB)
        new_mems = [] if not train else None
        st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d,
                  o_bs=T * d, o_rs=d)
        for l in range(L):
            s = l if train else 0
            h_in = ws.h[l] if train else ws.h[l & 1]
            h_out = ws.h[l + 1] if train else ws.h[(l + 1) & 1]
            if has_mem:
                cat = ws.cat[s]
                cat[:, :M].copy_(mems[l])
                cat[:, M:].copy_(h_in.view(B, T, d))
                x_qkv = cat.view(B * Kc, d)
            else:
                x_qkv = h_in
            if new_mems is not None:
                if has_mem:
                    new_mems.append(cat[:, Kc - M:].clone())
                elif T >= M:
                    new_mems.append(h_in.view(B, T, d)[:, T - M:].clone())
                else:
                    nm = torch.zeros(B, M, d, device=self.dev, dtype=torch.bfloat16)
                    nm[:, M - T:].copy_(h_in.view(B, T, d))
                    new_mems.append(nm)
            qkv = ws.qkv[s]
            ops.gemm(x_qkv, self._lw(l, 'dec_attn.qkv_net.weight'), qkv.view(B * Kc, 3 * d), B * Kc, 3 * d, d)
            if kv_sink is not None:
                kv_sink(l, qkv)   # decode prefill: projected K/V rows go to the per-layer rings
            ops.gemm(ws.phi, self._lw(l, 'dec_attn.r_net.weight'), ws.rd[s], M, d, d)
            ops.relattn_fwd(qkv[:, Kc - T:, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], ws.rd[s],
                            self._lw(l, 'dec_attn.r_w_bias', self.P), self._lw(l, 'dec_attn.r_r_bias', self.P),
                            ws.av[s], ws.lse[s], oph=ws.oph[s].view(B, T, d) if (train and ws.oph is not None) else None,
                            mph=ws.mph[s] if (train and ws.oph is not None) else None,
                            oph_all=bool(train and getattr(ws, 'fused_bwd', False)),
                            ph_buf=ws.ph[s] if (train and ws.ph is not None) else None, **st)
            ops.gemm(ws.av[s], self._lw(l, 'dec_attn.o_net.weight'), ws.tmp, N, d, d)
            ops.ln_residual_fwd(ws.tmp, h_in, self._lw(l, 'dec_attn.layer_norm.weight', self.P),
                                self._lw(l, 'dec_attn.layer_norm.bias', self.P), ws.h1[s],
                                ws.z1[l] if train else None, ws.st1[l][0] if train else None,
                                ws.st1[l][1] if train else None, eps=c.layer_norm_epsilon, drop_p=p, seed=seed,
                                site=self._site(l, 0))
            fl = F.GEMM_BIAS | F.GEMM_RELU | (F.GEMM_DROPOUT if p > 0 else 0)
            save_bits = train and ws.rmask is not None
            ops.gemm(ws.h1[s], self._lw(l, 'pos_ff.CoreNet.0.weight'), ws.a[s], N, Fi, d,
                     flags=fl | (F.GEMM_SAVE_RELU_MASK if save_bits else 0), aux=ws.rmask[s] if save_bits else None,
                     bias=self._lw(l, 'pos_ff.CoreNet.0.bias', self.P), drop_p=p, seed=seed, site=self._site(l, 1))
            ops.gemm(ws.a[s], self._lw(l, 'pos_ff.CoreNet.3.weight'), ws.tmp, N, d, Fi, flags=F.GEMM_BIAS,
                     bias=self._lw(l, 'pos_ff.CoreNet.3.bias', self.P))
            ops.ln_residual_fwd(ws.tmp, ws.h1[s], self._lw(l, 'pos_ff.layer_norm.weight', self.P),
                                self._lw(l, 'pos_ff.layer_norm.bias', self.P), h_out,
                                ws.z2[l] if train else None, ws.st2[l][0] if train else None,
                                ws.st2[l][1] if train else None, eps=c.layer_norm_epsilon, drop_p=p, seed=seed,
                                site=self._site(l, 2))
        hid = ws.h[L] if train else ws.h[L & 1]
        if p > 0:
            ops.dropout(hid, ws.hid_d, p, seed=seed, site=self.SITE_FINAL)
            hid = ws.hid_d
        ws.hid = hid
        nrow = self.layout.n_head_rows
        head_w = self.W[:self.layout.head_rows_padded * d].view(self.layout.head_rows_padded, d)
        boff = self.layout.entries['crit.out_layers.0.bias'][0]
        head_b = self.P[boff:boff + nrow]
        out = dict(loss=None, losses=None, logprobs=None, mems=new_mems)
        if self.bucketed_head:
            if labels is not None:
                lab = labels.contiguous().clone()
                ops.label_guard(lab, c.eos_token_id)       # transformer_xl.py:176-182
                ws.labels = lab
                ws.acc.zero_()
                self._bucketed_nll_fwd(ws, hid, lab, B, T)
                out['losses'] = ws.nll
                out['loss'] = ws.acc[0] / ws.acc[1]
            if want_logprobs and (labels is None or not train):
                out['logprobs'] = self._chunked_logprobs(hid, N, head_w, head_b).view(B, T, V)
            self._last = ws
            return out
        ops.gemm(hid, head_w, ws.logits, N, nrow, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=head_b)
        if labels is not None:
            lab = labels.contiguous().clone()
            ops.label_guard(lab, c.eos_token_id)       # transformer_xl.py:176-182
            ws.labels = lab
            ws.acc.zero_()
            ops.adaptive_nll_fwd(ws.logits, lab, ws.nll, ws.hlse, ws.acc, B, T, V, cut)
            out['losses'] = ws.nll
            out['loss'] = ws.acc[0] / ws.acc[1]        # mean over non-zero per-token losses (:200); stays on device
        if want_logprobs and (labels is None or not train):
            lp = torch.empty(N, V, device=self.dev, dtype=torch.float32)
            ops.adaptive_logprob(ws.logits, lp, N, V, cut)
            out['logprobs'] = lp.view(B, T, V)
        self._last = ws
        return out

    # ------------------------------------------------------------------ large-vocabulary head (csrc/head_large.hip)
    HEAD_CHUNK_BYTES = 1 << 30          # fp32 logits of one chunk of tokens

    def _hb_scratch(self, N):
        """per-token arrays of the bucketed head + the packed head operand [E[:c1] ; cluster_weight ; 0] and its bias"""
        c, dev = self.cfg, self.dev
        hb = self._hb
        ncl, c1, d = len(c.cutoffs), c.cutoffs[0], c.d_model
        if hb is None or hb.N != N:
            hb = self._hb = _WS()
            hb.N = N
            i32 = dict(device=dev, dtype=torch.int32)
            f32 = dict(device=dev, dtype=torch.float32)
            hb.perm = torch.empty(ncl + 2, N, **i32)
            hb.counts = torch.zeros(ncl + 2, **i32)
            hb.tgt_head, hb.tgt_tail = torch.empty(N, **i32), torch.empty(N, **i32)
            hb.hlse, hb.hpick, hb.tlse, hb.tpick = (torch.zeros(N, **f32) for _ in range(4))
            hb.nll_tok = torch.empty(N, **f32)
            hb.ldh = _r8(c1 + ncl)
            hb.wp = torch.zeros(hb.ldh, d, device=dev, dtype=torch.bfloat16)
            hb.bp = torch.zeros(hb.ldh, **f32)
        return hb

    def _chunk_rows(self, n, cols):
        ch = max(256, (self.HEAD_CHUNK_BYTES // (4 * max(cols, 1))) // 256 * 256)
        return min(n, ch)

    def _pack_head(self, hb):
        c = self.cfg
        V, ncl, c1, d = c.vocab_size, len(c.cutoffs), c.cutoffs[0], c.d_model
        head_w = self.W[:self.layout.head_rows_padded * d].view(self.layout.head_rows_padded, d)
        boff = self.layout.entries['crit.out_layers.0.bias'][0]
        hb.wp[:c1].copy_(head_w[:c1]); hb.wp[c1:c1 + ncl].copy_(head_w[V:V + ncl])
        hb.bp[:c1].copy_(self.P[boff:boff + c1]); hb.bp[c1:c1 + ncl].copy_(self.P[boff + V:boff + V + ncl])

    def _bucketed_nll_fwd(self, ws, hid, lab, B, T):
        """upstream ProjectedAdaptiveLogSoftmax.forward(hidden, labels): head softmax for every token (chunks of tokens), tail
        softmax of cluster i for the tokens whose label lies in it (bucketed, gathered, chunked)."""
        c = self.cfg
        V, cut, d = c.vocab_size, tuple(c.cutoffs), c.d_model
        ncl, c1, N = len(cut), cut[0], B * T
        hb = self._hb_scratch(N)
        self._pack_head(hb)
        ops.cluster_bucket(lab, V, cut, hb.perm, hb.counts, hb.tgt_head, hb.tgt_tail)
        counts = hb.counts.cpu().tolist()          # the one host sync of this path (upstream: mask_i.nonzero())
        hb.counts_host = counts
        head_w = self.W[:self.layout.head_rows_padded * d].view(self.layout.head_rows_padded, d)
        boff = self.layout.entries['crit.out_layers.0.bias'][0]
        ch = self._chunk_rows(N, hb.ldh)
        hl = torch.empty(ch, hb.ldh, device=self.dev, dtype=torch.float32)
        for r0 in range(0, N, ch):
            n = min(ch, N - r0)
            ops.gemm(hid[r0:r0 + n], hb.wp, hl, n, c1 + ncl, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=hb.bp)
            ops.rows_lse_pick(hl, c1 + ncl, n, hb.tgt_head, hb.hlse, hb.hpick, row0=r0)
        del hl
        bounds = (0,) + cut + (V,)
        for i in range(1, ncl + 1):
            n_i, lo, S = counts[i], bounds[i], bounds[i + 1] - bounds[i]
            if n_i == 0:
                continue
            ldt = _r8(S)
            ch = self._chunk_rows(n_i, ldt)
            tl = torch.empty(ch, ldt, device=self.dev, dtype=torch.float32)
            hc = torch.empty(ch, d, device=self.dev, dtype=torch.bfloat16)
            for j0 in range(0, n_i, ch):
                n = min(ch, n_i - j0)
                idx = hb.perm[i, j0:j0 + n]
                ops.gather_rows(hid, idx, hc, n)
                ops.gemm(hc, head_w[lo:lo + S], tl, n, S, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=self.P[boff + lo:boff + lo + S])
                ops.rows_lse_pick(tl, S, n, hb.tgt_tail, hb.tlse, hb.tpick, rows_idx=idx)
            del tl, hc
        ops.bucket_nll_finish(hb.hlse, hb.hpick, hb.tlse, hb.tpick, hb.tgt_head, hb.tgt_tail, ws.nll, hb.nll_tok, ws.acc, B, T)

    def _bucketed_head_bwd(self, ws, dy, B, T, grad_scale):
        """gradients of the bucketed head: logits are recomputed chunk by chunk (one more GEMM per chunk instead of keeping them);
        the shortlist / cluster-column gradient travels as a two-term bf16 sum like the small-vocabulary head's"""
        c = self.cfg
        V, cut, d = c.vocab_size, tuple(c.cutoffs), c.d_model
        ncl, c1, N = len(cut), cut[0], B * T
        hb, G, hid = self._hb, self.G, ws.hid
        AT = F.GEMM_OUT_F32_ATOMIC
        nrow_p = self.layout.head_rows_padded
        head_w = self.W[:nrow_p * d].view(nrow_p, d)
        g_head_w = G[:nrow_p * d].view(nrow_p, d)
        boff = self.layout.entries['crit.out_layers.0.bias'][0]
        ch = self._chunk_rows(N, hb.ldh)
        hl = torch.empty(ch, hb.ldh, device=self.dev, dtype=torch.float32)
        dl = torch.empty(ch, hb.ldh, device=self.dev, dtype=torch.bfloat16)
        dl_lo = torch.empty_like(dl)
        for r0 in range(0, N, ch):
            n = min(ch, N - r0)
            x = hid[r0:r0 + n]
            ops.gemm(x, hb.wp, hl, n, c1 + ncl, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=hb.bp)
            ops.rows_softmax_grad(hl, c1 + ncl, n, hb.tgt_head, hb.hlse, hb.nll_tok, ws.acc, grad_scale, dl, dl_lo, row0=r0)
            for term in (dl_lo, dl):
                ops.colsum(term, G[boff:boff + c1], n, c1)
                ops.colsum(term[:, c1:], G[boff + V:boff + V + ncl], n, ncl, ld=hb.ldh)
                ops.gemm(term, x, g_head_w[:c1], c1, d, n, trans_a=True, trans_b=True, flags=AT, ksplits=self._ks(c1, d, n))
                ops.gemm(term[:, c1:], x, g_head_w[V:V + ncl], ncl, d, n, trans_a=True, trans_b=True, flags=AT, lda=hb.ldh,
                         ksplits=self._ks(ncl, d, n))
            ops.gemm(dl_lo, hb.wp, dy[r0:r0 + n], n, d, hb.ldh, trans_b=True)
            ops.gemm(dl, hb.wp, dy[r0:r0 + n], n, d, hb.ldh, trans_b=True, flags=F.GEMM_ADD_AUX, aux=dy[r0:r0 + n])
        del hl, dl, dl_lo
        bounds = (0,) + cut + (V,)
        for i in range(1, ncl + 1):
            n_i, lo, S = hb.counts_host[i], bounds[i], bounds[i + 1] - bounds[i]
            if n_i == 0:
                continue
            ldt = _r8(S)
            ch = self._chunk_rows(n_i, ldt)
            tl = torch.empty(ch, ldt, device=self.dev, dtype=torch.float32)
            dtl = torch.empty(ch, ldt, device=self.dev, dtype=torch.bfloat16)
            hc = torch.empty(ch, d, device=self.dev, dtype=torch.bfloat16)
            dhc = torch.empty(ch, d, device=self.dev, dtype=torch.bfloat16)
            for j0 in range(0, n_i, ch):
                n = min(ch, n_i - j0)
                idx = hb.perm[i, j0:j0 + n]
                ops.gather_rows(hid, idx, hc, n)
                ops.gemm(hc, head_w[lo:lo + S], tl, n, S, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=self.P[boff + lo:boff + lo + S])
                ops.rows_softmax_grad(tl, S, n, hb.tgt_tail, hb.tlse, hb.nll_tok, ws.acc, grad_scale, dtl, None, rows_idx=idx)
                ops.colsum(dtl, G[boff + lo:boff + lo + S], n, S)
                ops.gemm(dtl, hc, g_head_w[lo:lo + S], S, d, n, trans_a=True, trans_b=True, flags=AT, ksplits=self._ks(S, d, n))
                # K = the padded row length: the pad columns of dtl are zero and the weight rows they meet are finite
                ops.gemm(dtl, head_w[lo:lo + ldt], dhc, n, d, ldt, trans_b=True)
                ops.scatter_add_rows(dhc, idx, dy, n)
            del tl, dtl, hc, dhc

    def _chunked_logprobs(self, hid, N, head_w, head_b):
        """labels=None branch at a large vocabulary: the full (N, V) log-probabilities the caller asked for, computed over
        chunks of tokens so that only one chunk's (rows, V + clusters) logits exist at a time"""
        c = self.cfg
        V, cut, d = c.vocab_size, tuple(c.cutoffs), c.d_model
        nrow, nrow_p = self.layout.n_head_rows, self.layout.head_rows_padded
        lp = torch.empty(N, V, device=self.dev, dtype=torch.float32)
        ch = self._chunk_rows(N, nrow_p)
        lg = torch.empty(ch, nrow_p, device=self.dev, dtype=torch.float32)
        for r0 in range(0, N, ch):
            n = min(ch, N - r0)
            ops.gemm(hid[r0:r0 + n], head_w, lg, n, nrow, d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=head_b)
            ops.adaptive_logprob(lg, lp[r0:r0 + n], n, V, cut)
        return lp

    # ------------------------------------------------------------------ backward
    def zero_grad(self):
        if self.G is None:
            self.G = torch.zeros_like(self.P)
        else:
            self.G.zero_()

    @staticmethod
    def _ks(m, n, k=0):
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        if k >= 49152:
            # long contractions (per-GPU batch 32: 65536 tokens) amortise a slice's atomics over more K-steps, and shorter
            # slices in ~3 rounds balance better than one round of long ones (scripts/perf_dw_ks.py 65536: ffn1 459 -> 429 us,
            # ffn2 456 -> 436, qkv 351 -> 343 at 10-13 slices; the 768 x 768 gradient is best at 12: 122 us, 146 at 24).  Inside the
            # C3 step the gain is within noise (weight-gradient GEMM 271 -> 269 us on average over a step's 61 launches)
            return 12 if tiles <= 48 else max(1, min(24, 1440 // tiles))
        # fill the 512 resident workgroup slots (256 CUs x 2) in one round; measured (scripts/perf_dw_ks.py): small gradients take as many K-slices as fill the slots (up to 24: a 512 x 512 gradient 86 -> 68 us); beyond 512 workgroups a second round starts
        return max(1, min(24, 512 // max(tiles, 1)))

    def backward(self, grad_scale: float = 1.0, layer_done=None):
        """Gradients of `loss` from the last train-mode forward, accumulated (+=) into self.G.
        `layer_done(name_prefix, lo, hi)` is called once a contiguous slice [lo, hi) of G is final (for overlap of the
        data-parallel all-reduce with the rest of the backward)."""
        ws, c = self._last, self.cfg
        B, T, Kc, p, seed = ws.B, ws.T, ws.Kc, ws.p, ws.seed
        d, H, dh, L, M, Fi = c.d_model, c.n_head, c.d_head, c.n_layer, c.mem_len, c.d_inner
        V, cut = c.vocab_size, tuple(c.cutoffs)
        if self.WT is None:
            self._refresh_wt(allocate=True)
        N = B * T
        G = self.G
        nrow, nrow_p = self.layout.n_head_rows, self.layout.head_rows_padded
        AT = F.GEMM_OUT_F32_ATOMIC
        dscale = 1.0 / (1.0 - p) if p > 0 else 1.0

        def gw(l, suffix):
            return self._lw(l, suffix, G)

        # ---- head
        dy, dy2 = ws.dA, None
        if self.bucketed_head:
            self._bucketed_head_bwd(ws, dy, B, T, grad_scale)
        else:
            # the logit gradient travels as a two-term bf16 sum (mxl_adaptive_nll_bwd_split): every consumer runs once per term and
            # accumulates -- the small term first where the output is rounded (the input gradient), so that it is rounded once
            ops.adaptive_nll_bwd(ws.logits, ws.labels, ws.nll, ws.hlse, ws.acc, ws.dlogits, B, T, V, cut, grad_scale,
                                 dlogits_lo=ws.dlogits_lo)
            head_w = self.W[:nrow_p * d].view(nrow_p, d)
            g_head_w = G[:nrow_p * d].view(nrow_p, d)
            boff = self.layout.entries['crit.out_layers.0.bias'][0]
            for term in (ws.dlogits_lo, ws.dlogits):
                ops.colsum(term, G[boff:boff + nrow], N, nrow)
                ops.gemm(term, ws.hid, g_head_w, nrow_p, d, N, trans_a=True, trans_b=True, flags=AT,
                         ksplits=self._ks(nrow_p, d, N))
            ops.gemm(ws.dlogits_lo, self.WT['head'], dy, N, d, self._head_kp())
            ops.gemm(ws.dlogits, self.WT['head'], dy, N, d, self._head_kp(), flags=F.GEMM_ADD_AUX, aux=dy)
        if p > 0:
            ops.dropout(dy, dy, p, seed=seed, site=self.SITE_FINAL)
        st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d,
                  o_bs=T * d, o_rs=d)
        # r_net's weight gradient uses the positional table centred over the distance axis: sum_d dRd[d] is exactly zero (the
        # score gradients of a softmax row sum to zero, and every query sees exactly M distances), so the constant part of phi
        # only ever multiplies the bf16 rounding noise of dG; see mxl_center_columns_bf16
        ops.center_columns(ws.phi, ws.phi_c)
        for l in reversed(range(L)):
            h_in = ws.h[l]
            # LN2 backward: dres -> dC (into h1 via residual), dx -> dD (into f_out)
            ops.ln_residual_bwd(dy, dy2, ws.z2[l], ws.st2[l][0], ws.st2[l][1], self._lw(l, 'pos_ff.layer_norm.weight', self.P),
                                ws.dC, ws.dD, gw(l, 'pos_ff.layer_norm.weight'), gw(l, 'pos_ff.layer_norm.bias'),
                                drop_p=p, seed=seed, site=self._site(l, 2), dxsum=gw(l, 'pos_ff.CoreNet.3.bias'))
            # FFN2 (its bias gradient = the column sums of dD: accumulated by the LayerNorm backward above)
            ops.gemm(ws.dD, ws.a[l], gw(l, 'pos_ff.CoreNet.3.weight'), d, Fi, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(d, Fi, N))
            if ws.rmask is not None:
                ops.gemm(ws.dD, self._lwt(l, 'pos_ff.CoreNet.3.weight'), ws.dF, N, Fi, d, flags=F.GEMM_RELU_BWD_BITS,
                         aux=ws.rmask[l], alpha=dscale, colsum=gw(l, 'pos_ff.CoreNet.0.bias'))
            else:
                ops.gemm(ws.dD, self._lwt(l, 'pos_ff.CoreNet.3.weight'), ws.dF, N, Fi, d, flags=F.GEMM_RELU_BWD,
                         aux=ws.a[l], alpha=dscale, colsum=gw(l, 'pos_ff.CoreNet.0.bias'))
            # FFN1 (its bias gradient = the column sums of dF: formed in the epilogue of the GEMM above)
            ops.gemm(ws.dF, ws.h1[l], gw(l, 'pos_ff.CoreNet.0.weight'), Fi, d, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(Fi, d, N))
            ops.gemm(ws.dF, self._lwt(l, 'pos_ff.CoreNet.0.weight'), ws.dD, N, d, Fi)
            # LN1 backward with both streams into h1: dC (residual) + dD (FFN1 dX)
            ops.ln_residual_bwd(ws.dC, ws.dD, ws.z1[l], ws.st1[l][0], ws.st1[l][1],
                                self._lw(l, 'dec_attn.layer_norm.weight', self.P), ws.dA, ws.dB,
                                gw(l, 'dec_attn.layer_norm.weight'), gw(l, 'dec_attn.layer_norm.bias'), drop_p=p,
                                seed=seed, site=self._site(l, 0))
            # now dA = grad into h_in via residual, dB = grad into o_net output
            ops.gemm(ws.dB, ws.av[l], gw(l, 'dec_attn.o_net.weight'), d, d, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(d, d, N))
            # d attn_vec; on the fused attention backward its row term delta = sum_e d attn_vec . attn_vec rides in the same launch
            delta_ready = False
            if ws.fused_bwd and not os.environ.get('MXL_HEADDOT_OFF'):
                delta_ready = ops.gemm_headdot(ws.dB, self._lwt(l, 'dec_attn.o_net.weight'), ws.dC, N, d, d, ws.av[l], T, ws.delta)
            else:
                ops.gemm(ws.dB, self._lwt(l, 'dec_attn.o_net.weight'), ws.dC, N, d, d)
            qkv, dqkv = ws.qkv[l], ws.dqkv
            ws.d_rd.zero_()
            if ws.fused_bwd:
                ops.relattn_bwd_fused(qkv[:, Kc - T:, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], ws.rd[l],
                                      self._lw(l, 'dec_attn.r_w_bias', self.P), self._lw(l, 'dec_attn.r_r_bias', self.P), ws.av[l],
                                      ws.dC, ws.lse[l], ws.delta, dqkv[:, Kc - T:, :d], dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:],
                                      ws.d_rd, gw(l, 'dec_attn.r_w_bias'), gw(l, 'dec_attn.r_r_bias'), ws.dq_slabs, ws.qr,
                                      dq_bs=Kc * 3 * d, dq_rs=3 * d, dkv_bs=Kc * 3 * d, dkv_rs=3 * d,
                                      oph=ws.oph[l].view(B, T, d) if ws.oph is not None else None,
                                      mph=ws.mph[l] if ws.oph is not None else None,
                                      ph_buf=ws.ph[l] if ws.ph is not None else None, ph_ready=ws.ph is not None,
                                      delta_ready=delta_ready, **st)
            else:
                ops.relattn_bwd(qkv[:, Kc - T:, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], ws.rd[l],
                                self._lw(l, 'dec_attn.r_w_bias', self.P), self._lw(l, 'dec_attn.r_r_bias', self.P), ws.av[l],
                                ws.dC, ws.lse[l], ws.delta, dqkv[:, Kc - T:, :d], dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:],
                                ws.dg, gw(l, 'dec_attn.r_w_bias'), gw(l, 'dec_attn.r_r_bias'), dq_bs=Kc * 3 * d, dq_rs=3 * d,
                                dkv_bs=Kc * 3 * d, dkv_rs=3 * d, d_rd=ws.d_rd, qr_buf=ws.qr,
                                oph=ws.oph[l].view(B, T, d) if ws.oph is not None else None,
                                mph=ws.mph[l] if ws.oph is not None else None, **st)
            # r_net: dW_r = d_rd^T . phi
            ops.cast_bf16(ws.d_rd, ws.d_rd16)
            ops.gemm(ws.d_rd16, ws.phi_c, gw(l, 'dec_attn.r_net.weight'), d, d, M, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(d, d))
            # qkv_net
            if ws.has_mem:
                dqkv[:, :M, :d].zero_()   # memory rows carry no query gradient
                x_qkv = ws.cat[l].view(B * Kc, d)
            else:
                x_qkv = h_in
            ops.gemm(dqkv.view(B * Kc, 3 * d), x_qkv, gw(l, 'dec_attn.qkv_net.weight'), 3 * d, d, B * Kc, trans_a=True,
                     trans_b=True, flags=AT, ksplits=self._ks(3 * d, d, B * Kc))
            if ws.has_mem:
                # only the current rows receive gradient (mems are detached): per-batch GEMM on the last T rows
                ops.gemm_batched(dqkv[:, M:], self._lw(l, 'dec_attn.qkv_net.weight'), ws.dB, T, d, 3 * d, lda=3 * d,
                                 ldb=d, ldc=d, trans_b=True, batch=B, bdiv=1, sA=(Kc * 3 * d, 0), sB=(0, 0), sC=(T * d, 0))
            else:
                ops.gemm(dqkv.view(N, 3 * d), self._lwt(l, 'dec_attn.qkv_net.weight'), ws.dB, N, d, 3 * d)
            dy, dy2 = ws.dA, ws.dB
            if layer_done is not None:
                layer_done(l)
        E_g = self.layout.view(G, 'transformer.word_emb.emb_layers.0.weight')
        ops.embed_bwd(ws.ids, dy, E_g, math.sqrt(d), drop_p=p, seed=seed, site=self.SITE_EMB, dout2=dy2)

    # ------------------------------------------------------------------ optimiser
    def optimizer_step(self, lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0,
                       grad_scale: float = 1.0):
        if self.m is None:
            self.m = torch.zeros_like(self.P)
            self.v = torch.zeros_like(self.P)
        self.step_count += 1
        self.rng_step += 1
        self._sumsq.zero_()
        if max_grad_norm and max_grad_norm > 0:
            ops.sumsq(self.G, self._sumsq)
        ops.adamw_step(self.P, self.G, self.m, self.v, self.W, self.layout.n_decay, lr, betas[0], betas[1], eps,
                       weight_decay, self.step_count, self._sumsq if max_grad_norm else None, max_grad_norm or 0.0,
                       grad_scale)
        self._refresh_wt()

    def grad_norm(self) -> torch.Tensor:
        return self._sumsq.sqrt()
