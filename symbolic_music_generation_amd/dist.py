"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in CPU tests).  The only collective on the training path is the gradient all-reduce; the reference itself never
ran multi-GPU (HF Trainer would have wrapped the model in DDP: SURVEY 2.2).

Buckets are the engine's natural contiguous slices of the flat fp32 gradient buffer -- one per layer (~28 MB at
12L/768d) -- issued asynchronously as soon as that layer's backward has been enqueued, so the exchange of layer l
overlaps the backward of layers l-1..0.  xGMI is point-to-point (7 links/GPU): a few large messages beat many small ones,
hence no finer bucketing.  Averaging (1/world) is folded into the fused AdamW kernel (grad_scale).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


_FORCE = os.environ.get('MXL_DIST_FORCE') == '1'      # exercise the collective path on a single rank (hardware smoke test)


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def layer_buckets(layout, n_layer: int) -> Tuple[List[List[Tuple[int, int]]], List[Tuple[int, int]]]:
    """Per-layer [(lo, hi), ...] slices of the flat buffer (decay-segment weights + no-decay-segment small params) and
    the remaining head/embedding slices (final only after the embedding backward)."""
    ent = layout.entries
    names = list(ent.keys())

    def span(prefix):
        offs = [(ent[n][0], ent[n][0] + _numel(ent[n][1])) for n in names if n.startswith(prefix)]
        return offs

    per_layer = []
    covered = []
    prefix = getattr(layout, 'layer_prefix', lambda l: f'transformer.layers.{l}.')
    for l in range(n_layer):
        offs = span(prefix(l))
        dec = [o for o in offs if o[0] < layout.n_decay]
        nod = [o for o in offs if o[0] >= layout.n_decay]
        sl = [(min(o[0] for o in dec), _r8(max(o[1] for o in dec))), (min(o[0] for o in nod), _r8(max(o[1] for o in nod)))]
        per_layer.append(sl)
        covered += sl
    covered.sort()
    rest, pos = [], 0
    for lo, hi in covered:
        if lo > pos:
            rest.append((pos, lo))
        pos = max(pos, hi)
    if pos < layout.total:
        rest.append((pos, layout.total))
    return per_layer, rest


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _r8(n):
    return (n + 7) // 8 * 8


class GradSync:
    """Overlapped all-reduce (SUM) of the engine's gradient buffer."""

    def __init__(self, engine):
        self.engine = engine
        n_layer = getattr(engine.cfg, 'n_layer', None) or len(engine.cfg.attn_layers)
        self.per_layer, self.rest = layer_buckets(engine.layout, n_layer)
        self.pending = []

    def layer_done(self, l: int):
        if not is_dist():
            return
        for lo, hi in self.per_layer[l]:
            self.pending.append(dist.all_reduce(self.engine.G[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        if not is_dist():
            return
        for lo, hi in self.rest:
            self.pending.append(dist.all_reduce(self.engine.G[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        for w in self.pending:
            w.wait()
        self.pending = []
