"""Reformer execution engine: same design as `xl_engine.XLEngine` -- flat fp32/bf16 parameter buffers with
HF state-dict names, explicit layer loop over libmusicxl kernels, saved activations instead of reversible recomputation
(identical arithmetic to HF's `_ReversibleFunction`; 288 GB of HBM make the recompute unnecessary).

    x1 = x2 = drop(word_emb) + axial_pos
    per layer:  y1 = x1 + drop(Wo . Attn(LN(x2)))        Attn = local chunked | LSH (hash -> stable sort -> chunked -> merge)
                y2 = x2 + drop(W2 . relu(drop(W1 . LN(y1) + b1)) + b2)
    logits = LN_2d(cat(y1, y2)) . Wd^T + b ;  loss = shifted cross-entropy (ignore -100)
"""
import math
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import ops
from ._lib import MusicXLError

F = ops


def _r8(n):
    return (n + 7) // 8 * 8


def auto_num_buckets(seq_len: int, chunk: int, max_pos: int):
    """LSHSelfAttention._set_num_buckets (HF515:791-809)"""
    p2 = (2 * (seq_len // chunk)).bit_length() - 1
    nb = 2 ** p2
    limit = 2 * max(int((max_pos // chunk) ** 0.5), chunk)
    if nb > limit:
        return [2 ** (p2 // 2), 2 ** (p2 - p2 // 2)]
    return nb


class RFLayout:
    """[decay | no-decay]; q,k,v (or qk,v) weights of a layer are adjacent so the projections are ONE GEMM."""

    def __init__(self, cfg):
        d, Fi, V = cfg.hidden_size, cfg.feed_forward_size, cfg.vocab_size
        hd = cfg.num_attention_heads * cfg.attention_head_size
        A0, A1 = cfg.axial_pos_shape
        d0, d1 = cfg.axial_pos_embds_dim
        self.head_rows_padded = _r8(V)
        decay = [('reformer.embeddings.word_embeddings.weight', (V, d)),
                 ('reformer.embeddings.position_embeddings.weights.0', (A0, 1, d0)),
                 ('reformer.embeddings.position_embeddings.weights.1', (1, A1, d1)),
                 ('lm_head.decoder.weight', (V, 2 * d))]
        if self.head_rows_padded > V:
            decay.append(('_pad.head_rows', (self.head_rows_padded - V, 2 * d)))
        nodecay = [('lm_head.bias', (V,)), ('reformer.encoder.layer_norm.weight', (2 * d,)),
                   ('reformer.encoder.layer_norm.bias', (2 * d,))]
        self.contig = {'lm_head.decoder.weight'}
        for i, kind in enumerate(cfg.attn_layers):
            p = f'reformer.encoder.layers.{i}.'
            names = ('query', 'key', 'value') if kind == 'local' else ('query_key', 'value')
            for n in names:
                decay.append((p + f'attention.self_attention.{n}.weight', (hd, d)))
            decay += [(p + 'attention.output.dense.weight', (d, hd)), (p + 'feed_forward.dense.dense.weight', (Fi, d)),
                      (p + 'feed_forward.output.dense.weight', (d, Fi))]
            nodecay += [(p + 'attention.layer_norm.weight', (d,)), (p + 'attention.layer_norm.bias', (d,)),
                        (p + 'feed_forward.layer_norm.weight', (d,)), (p + 'feed_forward.layer_norm.bias', (d,)),
                        (p + 'feed_forward.dense.dense.bias', (Fi,)), (p + 'feed_forward.output.dense.bias', (d,))]
        self.entries: "OrderedDict[str, tuple]" = OrderedDict()
        off = 0
        for name, shape in decay:
            self.entries[name] = (off, shape)
            n = math.prod(shape)
            off = off + n if name in self.contig and self.head_rows_padded > V else _r8(off + n)
        off = _r8(off)
        self.n_decay = off
        for name, shape in nodecay:
            self.entries[name] = (off, shape)
            off = _r8(off + math.prod(shape))
        self.total = _r8(off)

    def view(self, buf, name):
        off, shape = self.entries[name]
        return buf[off:off + math.prod(shape)].view(*shape)

    def real_names(self):
        return [n for n in self.entries if not n.startswith('_pad.')]

    @staticmethod
    def layer_prefix(l):
        return f'reformer.encoder.layers.{l}.'


class _WS:
    pass


class RFEngine:
    SITE_EMB, SITE_POS, SITE_FINAL = 0, 1, 2

    def __init__(self, cfg, device, seed: int = 77):
        if not torch.cuda.is_available():
            raise MusicXLError('RFEngine needs a GPU: the product path has no CPU fallback')
        self.cfg, self.dev = cfg, torch.device(device)
        assert cfg.attention_head_size in (16, 32, 64) and cfg.hidden_size % 8 == 0
        assert cfg.lsh_attn_chunk_length == 64 and cfg.local_attn_chunk_length == 64
        assert cfg.lsh_num_chunks_before == 1 and cfg.lsh_num_chunks_after == 0
        assert cfg.local_num_chunks_before == 1 and cfg.local_num_chunks_after == 0 and cfg.is_decoder
        self.layout = RFLayout(cfg)
        n = self.layout.total
        self.P = torch.zeros(n, device=self.dev, dtype=torch.float32)
        self.W = torch.zeros(n, device=self.dev, dtype=torch.bfloat16)
        self.G = self.m = self.v = None
        self.WT = None                      # [in][out] weight copies for the dX GEMMs (training only; see XLEngine)
        self.step_count, self.base_seed = 0, seed
        self.rng_step = 0          # dropout-mask / LSH-rotation stream position (see XLEngine)
        self._ws: Dict = {}
        self._sumsq = torch.zeros(1, device=self.dev)
        self.num_buckets = cfg.num_buckets
        self.last_buckets: Dict[int, torch.Tensor] = {}
        self.keep_buckets = False          # also keep the bucket ids of train-mode forwards (parity tests)
        self.init_weights(seed)

    # ---- params (same interface as XLEngine)
    def p32(self, name):
        return self.layout.view(self.P, name)

    def w16(self, name):
        return self.layout.view(self.W, name)

    def g32(self, name):
        return self.layout.view(self.G, name)

    def _l(self, l, suffix, buf=None):
        return self.layout.view(self.W if buf is None else buf, f'reformer.encoder.layers.{l}.{suffix}')

    def _proj_w(self, l, kind, buf=None):
        """stacked [Wq;Wk;Wv] (local) or [Wqk;Wv] (lsh) as one (n*hd, d) matrix"""
        first = 'query' if kind == 'local' else 'query_key'
        n = 3 if kind == 'local' else 2
        off, shape = self.layout.entries[f'reformer.encoder.layers.{l}.attention.self_attention.{first}.weight']
        b = self.W if buf is None else buf
        return b[off:off + n * shape[0] * shape[1]].view(n * shape[0], shape[1])

    def init_weights(self, seed):
        g = torch.Generator().manual_seed(seed)
        c = self.cfg
        host = torch.zeros(self.layout.total)
        for name in self.layout.real_names():
            off, shape = self.layout.entries[name]
            n = math.prod(shape)
            if 'position_embeddings.weights' in name:
                host[off:off + n] = c.axial_norm_std * torch.randn(n, generator=g)
            elif name.endswith('layer_norm.weight'):
                host[off:off + n] = 1.0
            elif name.endswith('bias'):
                pass
            else:
                host[off:off + n] = c.initializer_range * torch.randn(n, generator=g)
        self.P.copy_(host)
        self.sync_weights()

    def sync_weights(self):
        ops.cast_bf16(self.P, self.W)
        self._refresh_wt()

    def _wt_sources(self, l):
        kind = self.cfg.attn_layers[l]
        return {'proj': self._proj_w(l, kind), 'o': self._l(l, 'attention.output.dense.weight'),
                'ff1': self._l(l, 'feed_forward.dense.dense.weight'), 'ff2': self._l(l, 'feed_forward.output.dense.weight')}

    def _refresh_wt(self, allocate: bool = False):
        if self.WT is None and not allocate:
            return
        first = self.WT is None
        if first:
            self.WT = {}
            self._wt_groups = []
            # layers whose copy of one weight has the same shape and sits at a constant stride in the flat buffer (every other layer
            # when local and LSH layers alternate) are transposed by ONE batched launch: 8 launches per step instead of 24 at C4
            L = len(self.cfg.attn_layers)
            srcs = [self._wt_sources(l) for l in range(L)]
            base = self.W.data_ptr()
            for key in srcs[0]:
                by_shape: Dict = {}
                for l in range(L):
                    by_shape.setdefault(tuple(srcs[l][key].shape), []).append(l)
                for (o, i), layers in by_shape.items():
                    offs = [(srcs[l][key].data_ptr() - base) // 2 for l in layers]
                    stride = offs[1] - offs[0] if len(layers) > 1 else 0
                    if len(layers) > 1 and all(offs[j + 1] - offs[j] == stride for j in range(len(layers) - 1)) and stride > 0:
                        runs = [layers]
                    else:
                        runs, stride = [[l] for l in layers], 0
                    for run in runs:
                        stack = torch.empty(len(run), i, o, device=self.dev, dtype=torch.bfloat16)
                        for j, l in enumerate(run):
                            self.WT[(l, key)] = stack[j]
                        self._wt_groups.append((key, run[0], stack, o, i, len(run), stride))
        for key, l0, stack, o, i, n, stride in self._wt_groups:
            w = self._wt_sources(l0)[key]
            ops.transpose(w, stack, o, i, batch=n, src_bstride=stride, dst_bstride=i * o)
        nrow_p, d2 = self.layout.head_rows_padded, 2 * self.cfg.hidden_size
        if 'head' not in self.WT:
            self.WT['head'] = torch.zeros(d2, self._head_kp(), device=self.dev, dtype=torch.bfloat16)
        off = self.layout.entries['lm_head.decoder.weight'][0]
        ops.transpose(self.W[off:off + nrow_p * d2].view(nrow_p, d2), self.WT['head'], nrow_p, d2, ld_dst=self._head_kp())

    def _head_kp(self):
        return (self.layout.head_rows_padded + 63) // 64 * 64

    def state_dict(self):
        return OrderedDict((n, self.p32(n).detach().cpu().clone()) for n in self.layout.real_names())

    def load_state_dict(self, sd, strict=True):
        missing = []
        for name in self.layout.real_names():
            if name in sd:
                self.p32(name).copy_(sd[name].to(torch.float32).reshape(self.layout.entries[name][1]))
            else:
                missing.append(name)
        if strict and missing:
            raise KeyError(f'missing parameters: {missing[:4]}')
        self.sync_weights()

    def num_parameters(self):
        return sum(math.prod(self.layout.entries[n][1]) for n in self.layout.real_names())

    # ---- workspace
    def _workspace(self, B, T, train):
        key = (B, T, train)
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
        d, Fi, H, L, n_h = c.hidden_size, c.feed_forward_size, c.num_attention_heads, len(c.attn_layers), c.num_hashes
        N = B * T
        bf = dict(device=dev, dtype=torch.bfloat16)
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        ws = _WS()
        keep = L if train else 1
        ws.x1 = [torch.empty(N, d, **bf) for _ in range(L + 1)] if train else [torch.empty(N, d, **bf) for _ in range(2)]
        ws.x2 = [torch.empty(N, d, **bf) for _ in range(L + 1)] if train else [torch.empty(N, d, **bf) for _ in range(2)]
        ws.hn = [torch.empty(N, d, **bf) for _ in range(keep)]
        ws.st1 = [torch.empty(2, N, **f32) for _ in range(keep)]
        ws.qkv = [torch.empty(N, 3 * d, **bf) for _ in range(keep)]
        ws.av = [torch.empty(N, d, **bf) for _ in range(keep)]
        ws.out_r = [torch.empty(B, n_h, T, d, **bf) for _ in range(keep)]
        ws.lse = [torch.empty(B, n_h, H, T, **f32) for _ in range(keep)]
        ws.spos = [torch.empty(B * H, n_h * T, **i32) for _ in range(keep)]
        ws.sidx = torch.empty(B * H, n_h * T, **i32)
        ws.buckets = torch.empty(B, H, n_h * T, **i32)
        ws.h2 = [torch.empty(N, d, **bf) for _ in range(keep)]
        ws.st2 = [torch.empty(2, N, **f32) for _ in range(keep)]
        ws.a = [torch.empty(N, Fi, **bf) for _ in range(keep)]
        mb = ops.gemm_relu_mask_bytes(N, Fi) if train else 0      # relu(+dropout) mask bits for the backward GEMM (xl_engine.py)
        ws.rmask = [torch.empty(mb, device=dev, dtype=torch.uint8) for _ in range(keep)] if mb else None
        ws.cat = torch.empty(N, 2 * d, **bf)
        ws.hcat = torch.empty(N, 2 * d, **bf)
        ws.hcat_d = torch.empty(N, 2 * d, **bf)
        ws.stf = torch.empty(2, N, **f32)
        ws.logits = torch.empty(N, self.layout.head_rows_padded, **f32)
        ws.nll = torch.empty(B, T - 1, **f32)
        ws.hlse = torch.empty(N, 2, **f32)
        ws.acc = torch.zeros(2, **f32)
        if train:
            ws.dlogits = torch.empty(N, self._head_kp(), **bf)       # (row stride = K granule of the large-tile GEMM: see xl_engine)
            ws.dcat = torch.empty(N, 2 * d, **bf)
            ws.g1 = torch.empty(N, d, **bf); ws.g2 = torch.empty(N, d, **bf)
            ws.t1 = torch.empty(N, d, **bf); ws.t2 = torch.empty(N, d, **bf)
            ws.dF = torch.empty(N, Fi, **bf)
            # one (T, d) slab per hash round: the chunk kernels write every element once, the key-normalisation backward sums the rounds
            ws.dq = torch.empty(B * n_h * T, d, **f32); ws.dk = torch.empty(B * n_h * T, d, **f32)
            ws.dv = torch.empty(B * n_h * T, d, **f32) if n_h > 1 else None
            ws.dqkv = torch.empty(N, 3 * d, **bf)
            ws.dout_r = torch.empty(B, n_h, T, d, **bf)
            ws.dlse = torch.empty(B, n_h, H, T, **f32)
        self._ws[key] = ws
        return ws

    def _site(self, l, k):
        return 8 + 8 * l + k

    def _factors(self, T):
        if self.num_buckets is None:
            self.num_buckets = auto_num_buckets(T, 64, self.cfg.max_position_embeddings)
        nb = self.num_buckets
        return [nb] if isinstance(nb, int) else list(nb)

    # ---- forward
    def forward(self, input_ids, labels=None, train=False, rotations: Optional[Dict[int, torch.Tensor]] = None,
                buckets_override: Optional[Dict[int, torch.Tensor]] = None, n_real: Optional[int] = None, layer_sink=None):
        """`n_real` < T: positions n_real.. are HF's eval-mode padding to a multiple of the chunk length (HF515:2012-2040): they
        hash to ONE extra bucket and the per-round bucket offsets widen to num_buckets + 1 (HF515:746-756).  `layer_sink(l, kind,
        qkv, buckets)` sees every layer's projections (N, nproj*d) and bucket ids right after they are computed (cached decoding
        fills its caches from the prompt pass this way)."""
        c = self.cfg
        B, T = input_ids.shape
        single = T <= 64          # HF's standard-attention case: no hashing, no sort, no look-back chunk (HF515:547-549)
        if not single and T % 64 != 0:
            raise NotImplementedError('HIP Reformer path: sequences longer than one chunk must be a multiple of the chunk '
                                      'length 64 (HF pads them in eval mode and refuses them in training)')
        A0, A1 = c.axial_pos_shape
        if train and A0 * A1 != T:
            raise ValueError(f'If training, prod(axial_pos_shape) {A0 * A1} must equal the sequence length {T} (HF515:231-238)')
        if T > A0 * A1:
            raise ValueError('sequence longer than max_position_embeddings')
        d, Fi, H, dh, n_h = c.hidden_size, c.feed_forward_size, c.num_attention_heads, c.attention_head_size, c.num_hashes
        if single:
            n_h = 1
        L, V = len(c.attn_layers), c.vocab_size
        N = B * T
        ws = self._workspace(B, T, train)
        ws.single = single
        p = float(c.hidden_dropout_prob) if train else 0.0
        p_loc = float(c.local_attention_probs_dropout_prob) if train else 0.0
        p_lsh = float(c.lsh_attention_probs_dropout_prob) if train else 0.0
        seed = ops.mix_seed(self.base_seed, self.rng_step)
        ids = input_ids.contiguous()
        ws.ids, ws.B, ws.T, ws.p, ws.p_loc, ws.p_lsh, ws.seed = ids, B, T, p, p_loc, p_lsh, seed
        d0 = c.axial_pos_embds_dim[0]
        W0 = self.p32('reformer.embeddings.position_embeddings.weights.0').view(A0, d0)
        W1 = self.p32('reformer.embeddings.position_embeddings.weights.1').view(A1, d - d0)
        ops.axial_embed_fwd(ids, self.w16('reformer.embeddings.word_embeddings.weight'), W0, W1, ws.x1[0].view(B, T, d), A0, A1,
                            drop_p=p, seed=seed, site_emb=self.SITE_EMB, site_pos=self.SITE_POS)
        ws.x2[0].copy_(ws.x1[0])
        factors = self._factors(T) if not single else [2]
        NB = math.prod(factors)
        ws.rot = {}
        for l, kind in enumerate(c.attn_layers):
            s = l if train else 0
            x1, x2 = (ws.x1[l], ws.x2[l]) if train else (ws.x1[l & 1], ws.x2[l & 1])
            y1, y2 = (ws.x1[l + 1], ws.x2[l + 1]) if train else (ws.x1[(l + 1) & 1], ws.x2[(l + 1) & 1])
            ops.ln_residual_fwd(x2, None, self._l(l, 'attention.layer_norm.weight', self.P),
                                self._l(l, 'attention.layer_norm.bias', self.P), ws.hn[s], None, ws.st1[s][0], ws.st1[s][1],
                                eps=c.layer_norm_eps)
            nproj = 3 if kind == 'local' else 2
            qkv = ws.qkv[s].view(-1)[:N * nproj * d].view(N, nproj * d)
            ops.gemm(ws.hn[s], self._proj_w(l, kind), qkv, N, nproj * d, d)
            rs, bs = nproj * d, T * nproj * d
            if layer_sink is not None and (kind == 'local' or single):
                layer_sink(l, kind, qkv, None)
            if kind == 'local':
                ops.chunk_attn_fwd(qkv, qkv[:, d:], qkv[:, 2 * d:], None, ws.av[s], ws.lse[s], B, T, H, dh, 1, 0, bs, rs,
                                   drop_p=p_loc, seed=seed, site=self._site(l, 0))
            elif single:
                ops.chunk_attn_fwd(qkv, qkv, qkv[:, d:], None, ws.av[s], ws.lse[s], B, T, H, dh, 1, 1, bs, rs,
                                   drop_p=p_lsh, seed=seed, site=self._site(l, 0))
            else:
                if buckets_override is not None and l in buckets_override:
                    ws.buckets.copy_(buckets_override[l].to(torch.int32).view(B, H, n_h * T))
                else:
                    if rotations is not None and l in rotations:
                        rot = rotations[l].to(self.dev, torch.float32).contiguous()
                    else:   # HF draws fresh rotations from the global RNG every forward (hash_seed=None in the reference)
                        g = torch.Generator(device=self.dev).manual_seed((seed * 131 + l) & 0x7FFFFFFFFFFFFFFF)
                        rot = torch.randn(H, dh, n_h, sum(factors) // 2, device=self.dev, generator=g)
                    ws.rot[l] = rot
                    ops.lsh_hash(qkv, bs, rs, rot, ws.buckets, B, T, H, dh, n_h, factors)
                    if n_real is not None and n_real < T:
                        ops.lsh_fix_buckets(ws.buckets, B * H, n_h, T, n_real, NB)
                padded = n_real is not None and n_real < T
                if layer_sink is not None:
                    layer_sink(l, kind, qkv, ws.buckets)
                self.last_buckets[l] = ws.buckets.clone() if (not train or self.keep_buckets) else None
                ops.lsh_sort(ws.buckets, ws.sidx, ws.spos[s], B * H, n_h * T, T, (NB + 1 if padded else NB) * n_h)
                tgt = ws.av[s] if n_h == 1 else ws.out_r[s]
                ops.chunk_attn_fwd(qkv, qkv, qkv[:, d:], ws.spos[s], tgt, ws.lse[s], B, T, H, dh, n_h, 1, bs, rs,
                                   drop_p=p_lsh, seed=seed, site=self._site(l, 0))
                if n_h > 1:
                    ops.lsh_combine(ws.out_r[s], ws.lse[s], ws.av[s], B, T, H, dh, n_h)
            fl = F.GEMM_ADD_AUX | (F.GEMM_DROPOUT if p > 0 else 0)
            ops.gemm(ws.av[s], self._l(l, 'attention.output.dense.weight'), y1, N, d, d, flags=fl, aux=x1, drop_p=p, seed=seed,
                     site=self._site(l, 1))
            ops.ln_residual_fwd(y1, None, self._l(l, 'feed_forward.layer_norm.weight', self.P),
                                self._l(l, 'feed_forward.layer_norm.bias', self.P), ws.h2[s], None, ws.st2[s][0], ws.st2[s][1],
                                eps=c.layer_norm_eps)
            fl = F.GEMM_BIAS | F.GEMM_RELU | (F.GEMM_DROPOUT if p > 0 else 0)   # relu(drop(x)) == drop(relu(x))
            save_bits = train and ws.rmask is not None
            ops.gemm(ws.h2[s], self._l(l, 'feed_forward.dense.dense.weight'), ws.a[s], N, Fi, d,
                     flags=fl | (F.GEMM_SAVE_RELU_MASK if save_bits else 0), aux=ws.rmask[s] if save_bits else None,
                     bias=self._l(l, 'feed_forward.dense.dense.bias', self.P), drop_p=p, seed=seed, site=self._site(l, 2))
            fl = F.GEMM_BIAS | F.GEMM_ADD_AUX | (F.GEMM_DROPOUT if p > 0 else 0)
            ops.gemm(ws.a[s], self._l(l, 'feed_forward.output.dense.weight'), y2, N, d, Fi, flags=fl,
                     bias=self._l(l, 'feed_forward.output.dense.bias', self.P), aux=x2, drop_p=p, seed=seed,
                     site=self._site(l, 3))
        xL1, xL2 = (ws.x1[L], ws.x2[L]) if train else (ws.x1[L & 1], ws.x2[L & 1])
        ws.cat[:, :d].copy_(xL1)
        ws.cat[:, d:].copy_(xL2)
        ops.ln_residual_fwd(ws.cat, None, self.p32('reformer.encoder.layer_norm.weight'), self.p32('reformer.encoder.layer_norm.bias'),
                            ws.hcat, None, ws.stf[0], ws.stf[1], eps=c.layer_norm_eps)
        hid = ws.hcat
        if p > 0:
            ops.dropout(ws.hcat, ws.hcat_d, p, seed=seed, site=self.SITE_FINAL)
            hid = ws.hcat_d
        ws.hid = hid
        off = self.layout.entries['lm_head.decoder.weight'][0]
        nrow_p = self.layout.head_rows_padded
        head_w = self.W[off:off + nrow_p * 2 * d].view(nrow_p, 2 * d)
        ops.gemm(hid, head_w, ws.logits, N, V, 2 * d, flags=F.GEMM_OUT_F32 | F.GEMM_BIAS, bias=self.p32('lm_head.bias'))
        out = dict(loss=None, logits=ws.logits[:, :V].view(B, T, V))
        if labels is not None:
            ws.labels = labels.contiguous()
            ws.acc.zero_()
            ops.adaptive_nll_fwd(ws.logits, ws.labels, ws.nll, ws.hlse, ws.acc, B, T, V, ())
            out['loss'] = ws.acc[0] / ws.acc[1]
        self._last = ws
        return out

    # ---- backward
    def zero_grad(self):
        if self.G is None:
            self.G = torch.zeros_like(self.P)
        else:
            self.G.zero_()

    @staticmethod
    def _ks(m, n):
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        # fill the 512 resident workgroup slots (256 CUs x 2) in one round; measured (scripts/perf_dw_ks.py): small gradients take as many K-slices as fill the slots (up to 24); beyond 512 workgroups a second round starts
        return max(1, min(24, 512 // max(tiles, 1)))

    def backward(self, grad_scale=1.0, layer_done=None):
        ws, c = self._last, self.cfg
        B, T, p, seed = ws.B, ws.T, ws.p, ws.seed
        d, Fi, H, dh, n_h = c.hidden_size, c.feed_forward_size, c.num_attention_heads, c.attention_head_size, c.num_hashes
        if getattr(ws, 'single', False):
            n_h = 1
        L, V = len(c.attn_layers), c.vocab_size
        N = B * T
        if self.WT is None:
            self._refresh_wt(allocate=True)
        G, AT = self.G, F.GEMM_OUT_F32_ATOMIC
        dscale = 1.0 / (1.0 - p) if p > 0 else 1.0
        gl = lambda l, sfx: self._l(l, sfx, G)
        nrow_p = self.layout.head_rows_padded
        ops.adaptive_nll_bwd(ws.logits, ws.labels, ws.nll, ws.hlse, ws.acc, ws.dlogits, B, T, V, (), grad_scale)
        off = self.layout.entries['lm_head.decoder.weight'][0]
        head_w = self.W[off:off + nrow_p * 2 * d].view(nrow_p, 2 * d)
        g_head = G[off:off + nrow_p * 2 * d].view(nrow_p, 2 * d)
        ops.colsum(ws.dlogits, self.g32('lm_head.bias'), N, V)
        ops.gemm(ws.dlogits, ws.hid, g_head, nrow_p, 2 * d, N, trans_a=True, trans_b=True, flags=AT, ksplits=self._ks(nrow_p, 2 * d))
        ops.gemm(ws.dlogits, self.WT['head'], ws.dcat, N, 2 * d, self._head_kp())
        if p > 0:
            ops.dropout(ws.dcat, ws.dcat, p, seed=seed, site=self.SITE_FINAL)
        ops.ln_residual_bwd(ws.dcat, None, ws.cat, ws.stf[0], ws.stf[1], self.p32('reformer.encoder.layer_norm.weight'),
                            ws.hcat, None, self.g32('reformer.encoder.layer_norm.weight'), self.g32('reformer.encoder.layer_norm.bias'))
        # ws.hcat now holds d cat; split into the two streams
        ws.g1.copy_(ws.hcat[:, :d]); ws.g2.copy_(ws.hcat[:, d:])
        g1, g2, t1, t2 = ws.g1, ws.g2, ws.t1, ws.t2
        # the masked copies of g2 / g1 that the two branch inputs need (the forward's dropout masks regenerated) come out of the
        # LayerNorm backward that forms g2 / g1 -- mxl_ln_residual_bwd_add_drop -- instead of a pass of their own
        # (MXL_RF_NO_LN_DROP=1: the separate passes, for A/B and the equivalence test)
        fuse_drop = p > 0 and d <= 1024 and N * d < (1 << 32) and os.environ.get('MXL_RF_NO_LN_DROP') != '1'
        dff_ready = False
        for l in reversed(range(L)):
            kind = c.attn_layers[l]
            x1, x2, y1 = ws.x1[l], ws.x2[l], ws.x1[l + 1]
            # ---- y2 = x2 + drop(a W2^T + b2)
            if p > 0 and dff_ready:     # t1 = dropout(g2) and its column sums: written by the layer above's last LayerNorm backward
                dff = t1
            elif p > 0:     # the forward's mask regenerated and the bias gradient (column sums of the masked gradient) in one pass
                ops.dropout_colsum(g2, t1, gl(l, 'feed_forward.output.dense.bias'), N, d, p, seed, self._site(l, 3))
                dff = t1
            else:
                dff = g2
                ops.colsum(dff, gl(l, 'feed_forward.output.dense.bias'), N, d)
            ops.gemm(dff, ws.a[l], gl(l, 'feed_forward.output.dense.weight'), d, Fi, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(d, Fi))
            if ws.rmask is not None:
                ops.gemm(dff, self.WT[(l, 'ff2')], ws.dF, N, Fi, d, flags=F.GEMM_RELU_BWD_BITS,
                         aux=ws.rmask[l], alpha=dscale, colsum=gl(l, 'feed_forward.dense.dense.bias'))
            else:
                ops.gemm(dff, self.WT[(l, 'ff2')], ws.dF, N, Fi, d, flags=F.GEMM_RELU_BWD,
                         aux=ws.a[l], alpha=dscale, colsum=gl(l, 'feed_forward.dense.dense.bias'))
            ops.gemm(ws.dF, ws.h2[l], gl(l, 'feed_forward.dense.dense.weight'), Fi, d, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(Fi, d))
            ops.gemm(ws.dF, self.WT[(l, 'ff1')], t2, N, d, Fi)
            # g1 <- g1 + LN2-backward(t2)            (y1 feeds the FF branch and the y1 output)
            if fuse_drop:       # ... and t2 (its own input) <- dropout(new g1) under the attention-output site
                ops.ln_bwd_add_drop(t2, None, y1, ws.st2[l][0], ws.st2[l][1], self._l(l, 'feed_forward.layer_norm.weight', self.P), g1,
                                    t1, t2, None, gl(l, 'feed_forward.layer_norm.weight'), gl(l, 'feed_forward.layer_norm.bias'),
                                    p, seed, self._site(l, 1))
            else:
                ops.ln_bwd_add(t2, None, y1, ws.st2[l][0], ws.st2[l][1], self._l(l, 'feed_forward.layer_norm.weight', self.P), g1, t1,
                               gl(l, 'feed_forward.layer_norm.weight'), gl(l, 'feed_forward.layer_norm.bias'))
            g1, t1 = t1, g1
            # ---- y1 = x1 + drop(av Wo^T)
            if fuse_drop:
                dao = t2
            elif p > 0:
                ops.dropout(g1, t2, p, seed=seed, site=self._site(l, 1))
                dao = t2
            else:
                dao = g1
            ops.gemm(dao, ws.av[l], gl(l, 'attention.output.dense.weight'), d, d, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(d, d))
            dav = ws.dF.view(-1)[:N * d].view(N, d)          # scratch
            ops.gemm(dao, self.WT[(l, 'o')], dav, N, d, d)
            nproj = 3 if kind == 'local' else 2
            qkv = ws.qkv[l].view(-1)[:N * nproj * d].view(N, nproj * d)
            dqkv = ws.dqkv.view(-1)[:N * nproj * d].view(N, nproj * d)
            rs, bs = nproj * d, T * nproj * d
            if kind == 'local':
                # one round: every gradient element is written once, in bf16, straight into the (N, 3d) GEMM operand
                ops.chunk_attn_bwd(qkv, qkv[:, d:], qkv[:, 2 * d:], None, ws.av[l], ws.lse[l], dav, None, None, None, None,
                                   B, T, H, dh, 1, 0, bs, rs, drop_p=ws.p_loc, seed=seed, site=self._site(l, 0),
                                   dq16=dqkv, dk16=dqkv[:, d:], dv16=dqkv[:, 2 * d:], ld16=3 * d)
            else:
                if n_h > 1:
                    ops.lsh_combine_bwd(ws.out_r[l], ws.lse[l], ws.av[l], dav, ws.dout_r, ws.dlse, B, T, H, dh, n_h)
                    o_in, do_in, dl_in = ws.out_r[l], ws.dout_r, ws.dlse
                else:
                    o_in, do_in, dl_in = ws.av[l], dav, None
                ops.chunk_attn_bwd(qkv, qkv, qkv[:, d:], None if getattr(ws, 'single', False) else ws.spos[l], o_in, ws.lse[l], do_in,
                                   dl_in, ws.dq, ws.dk, ws.dv if n_h > 1 else None, B, T, H,
                                   dh, n_h, 1, bs, rs, drop_p=ws.p_lsh, seed=seed, site=self._site(l, 0),
                                   dv16=None if n_h > 1 else dqkv[:, d:], ld16=0 if n_h > 1 else 2 * d)
                if n_h > 1:     # the rounds' slabs summed on the way in; the value gradient's sum goes out as bf16 beside dqk
                    ops.lsh_keynorm_bwd_rounds(qkv, bs, rs, ws.dq, ws.dk, ws.dv, dqkv, dqkv[:, d:], B, T, H, dh, n_h, ld_dqk=2 * d, ld_dv=2 * d)
                else:
                    ops.lsh_keynorm_bwd(qkv, bs, rs, ws.dq, ws.dk, dqkv, B, T, H, dh, ld_dqk=2 * d)
            ops.gemm(dqkv, ws.hn[l], self._proj_w(l, kind, G), nproj * d, d, N, trans_a=True, trans_b=True, flags=AT,
                     ksplits=self._ks(nproj * d, d))
            dhn = ws.dF.view(-1)[:N * d].view(N, d)
            ops.gemm(dqkv, self.WT[(l, 'proj')], dhn, N, d, nproj * d)
            # g2 <- g2 + LN1-backward(dhn)
            if fuse_drop and l > 0:     # ... and t1 <- dropout(new g2) under the NEXT layer's FFN-output site, with that layer's bias gradient
                ops.ln_bwd_add_drop(dhn, None, x2, ws.st1[l][0], ws.st1[l][1], self._l(l, 'attention.layer_norm.weight', self.P), g2,
                                    t2, t1, gl(l - 1, 'feed_forward.output.dense.bias'), gl(l, 'attention.layer_norm.weight'),
                                    gl(l, 'attention.layer_norm.bias'), p, seed, self._site(l - 1, 3))
                dff_ready = True
            else:
                ops.ln_bwd_add(dhn, None, x2, ws.st1[l][0], ws.st1[l][1], self._l(l, 'attention.layer_norm.weight', self.P), g2, t2,
                               gl(l, 'attention.layer_norm.weight'), gl(l, 'attention.layer_norm.bias'))
                dff_ready = False
            g2, t2 = t2, g2
            if layer_done is not None:
                layer_done(l)
        A0, A1 = c.axial_pos_shape
        d0 = c.axial_pos_embds_dim[0]
        ops.axial_embed_bwd(ws.ids, g1, self.g32('reformer.embeddings.word_embeddings.weight'),
                            self.g32('reformer.embeddings.position_embeddings.weights.0').view(A0, d0),
                            self.g32('reformer.embeddings.position_embeddings.weights.1').view(A1, d - d0), A0, A1, drop_p=p,
                            seed=seed, site_emb=self.SITE_EMB, site_pos=self.SITE_POS, dout2=g2)

    def optimizer_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0, grad_scale=1.0):
        if self.m is None:
            self.m = torch.zeros_like(self.P)
            self.v = torch.zeros_like(self.P)
        self.step_count += 1
        self.rng_step += 1
        self._sumsq.zero_()
        if max_grad_norm and max_grad_norm > 0:
            ops.sumsq(self.G, self._sumsq)
        ops.adamw_step(self.P, self.G, self.m, self.v, self.W, self.layout.n_decay, lr, betas[0], betas[1], eps, weight_decay,
                       self.step_count, self._sumsq if max_grad_norm else None, max_grad_norm or 0.0, grad_scale)
        self._refresh_wt()

    def grad_norm(self):
        return self._sumsq.sqrt()
