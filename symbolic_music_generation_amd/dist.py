"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in CPU tests).  The only collective on the training path is the gradient all-reduce; the reference itself never
ran multi-GPU (HF Trainer would have wrapped the model in DDP: SURVEY 2.2).

Buckets are the engine's natural contiguous slices of the flat fp32 gradient buffer -- one per layer (~28 MB at
12L/768d) -- issued asynchronously as soon as that layer's backward has been enqueued, so the exchange of layer l
overlaps the backward of layers l-1..0.  xGMI is point-to-point (7 links/GPU): a few large messages beat many small ones,
hence no finer bucketing.  Averaging (1/world) is folded into the fused AdamW kernel (grad_scale).
"""
import os
import sys
import time
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


_FORCE = os.environ.get('MXL_DIST_FORCE') == '1'      # exercise the collective path on a single rank (hardware smoke test)
_TRACE = os.environ.get('MXL_DIST_TRACE') == '1'      # print the host time spent issuing / waiting for the all-reduces


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def layer_buckets(layout, n_layer: int) -> Tuple[List[List[Tuple[int, int]]], List[Tuple[int, int]]]:
    """Per-layer [(lo, hi)] slice of the flat buffer -- the layer's decay-segment weights, one contiguous ~28 MB message at
    12L/768d -- and the remainder, the TAIL bucket(s): head / embedding rows and the whole no-decay segment (every layer's
    LayerNorm parameters and biases: a few KB per layer, latency-bound as messages of their own, so they travel together in one
    message once the embedding backward is enqueued).  12L: 12 + 2 all-reduces per step (26 when each layer sent its own
    no-decay slice)."""
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
        sl = [(min(o[0] for o in dec), _r8(max(o[1] for o in dec)))]
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
    """Overlapped all-reduce (SUM) of the engine's gradient buffer.

    `dtype='fp32'` (the default, what HF's DDP exchanges in the reference stack) all-reduces the slices of the fp32 gradient
    buffer in place.  `dtype='bf16'` (opt-in: `GradSync(engine, 'bf16')`, the `grad_exchange_dtype` train argument, or
    `MXL_DP_DTYPE=bf16`): a bucket is narrowed into a bf16 staging buffer by a HIP kernel, all-reduced at half the bytes
    (186 MB per step at 12L/768d instead of 372 MB), and widened back into the fp32 gradient buffer once it has
    arrived -- the fp32 master weights, Adam moments and the clip norm never see bf16 storage; each rank's contribution is
    rounded to 8 significant bits and RCCL sums in bf16 (error bound: tests/test_dp_cpu.py::test_bf16_exchange_error_bound).  The transport is `torch.distributed` (backend "nccl" = RCCL over xGMI): the
    communicator, its bootstrap and its stream ordering against torch's allocator are torch's; libmusicxl owns the casts.
    """

    def __init__(self, engine, dtype: Optional[str] = None):
        self.engine = engine
        n_layer = getattr(engine.cfg, 'n_layer', None) or len(engine.cfg.attn_layers)
        self.per_layer, self.rest = layer_buckets(engine.layout, n_layer)
        self.pending = []
        self.dtype = (dtype or os.environ.get('MXL_DP_DTYPE') or 'fp32').lower()
        self.dtype = {'f32': 'fp32', 'float32': 'fp32', 'bfloat16': 'bf16'}.get(self.dtype, self.dtype)
        if self.dtype not in ('bf16', 'fp32'):
            raise ValueError(f'gradient exchange dtype {self.dtype!r}: bf16 or fp32')
        self._stage = {}
        self.host_s = {'issue': 0.0, 'wait': 0.0, 'calls': 0, 'steps': 0}
        # Compute units left free of the persistent GEMM grids while a collective may be in flight (MXL_RESERVE_CUS, default 0): those
        # grids hold every CU for a whole launch (~0.2-0.7 ms), so RCCL's reduction kernels -- issued on their own stream right after
        # a layer's backward is enqueued -- could not start beside them.  Only when there is a collective to overlap.
        self.reserved_cus = int(os.environ.get('MXL_RESERVE_CUS', '0') or 0)
        if self.reserved_cus and is_dist() and torch.cuda.is_available() and getattr(engine, 'dev', torch.device('cpu')).type == 'cuda':
            from . import ops
            ops.set_reserved_cus(self.reserved_cus)

    def _issue(self, lo: int, hi: int):
        G = self.engine.G
        if self.dtype == 'fp32' or not G.is_cuda:        # the narrowing / widening kernels are device code; host tensors (gloo
            self.pending.append((dist.all_reduce(G[lo:hi], op=dist.ReduceOp.SUM, async_op=True), None, lo, hi))   # tests) go as they are
            return
        from . import ops
        buf = self._stage.get((lo, hi))
        if buf is None:
            buf = self._stage[(lo, hi)] = torch.empty(hi - lo, device=G.device, dtype=torch.bfloat16)
        ops.cast_bf16(G[lo:hi], buf)
        self.pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), buf, lo, hi))

    def layer_done(self, l: int):
        if not is_dist():
            return
        t = time.perf_counter() if _TRACE else 0.0
        for lo, hi in self.per_layer[l]:
            self._issue(lo, hi)
        if _TRACE:
            self.host_s['issue'] += time.perf_counter() - t
            self.host_s['calls'] += len(self.per_layer[l])

    def finish(self):
        if not is_dist():
            return
        t = time.perf_counter() if _TRACE else 0.0
        for lo, hi in self.rest:
            self._issue(lo, hi)
        t1 = time.perf_counter() if _TRACE else 0.0
        for work, buf, lo, hi in self.pending:
            work.wait()
            if buf is not None:
                from . import ops
                ops.cast_f32(buf, self.engine.G[lo:hi])
        if _TRACE:
            self.host_s['issue'] += t1 - t
            self.host_s['calls'] += len(self.rest)
            self.host_s['wait'] += time.perf_counter() - t1
            self.host_s['steps'] += 1
            if self.host_s['steps'] % 5 == 0 and rank() == 0:
                n = self.host_s['steps']
                print(f"[GradSync] host time per step: issue {1e3 * self.host_s['issue'] / n:.2f} ms over {self.host_s['calls'] / n:.0f} "
                      f"all-reduce calls, wait {1e3 * self.host_s['wait'] / n:.2f} ms", file=sys.stderr, flush=True)
        self.pending = []
