"""`torch.nn.Module` face of the HIP engines: what makes the models drop-ins for the reference's HF-`Trainer` harness.

The reference drives its models as ordinary modules (musicnlp/util/train/train_util_wrap.py:88-144 and
musicnlp/trainer/train.py:350-367): `outputs = model(**inputs); loss = outputs.loss; loss.backward();
clip_grad_norm_(model.parameters(), 1.0); optimizer.step()`.  `EngineModule` provides exactly that over an engine whose
state is ONE flat fp32 buffer `P`:

* every parameter is an `nn.Parameter` that is a view into `P`, registered under upstream's dotted state-dict name in a
  tree of plain container modules, so `named_parameters()`, `state_dict()`, HF's weight-decay grouping by name and any
  `torch.optim` optimizer work unchanged;
* a train-mode forward returns a `loss` produced by a `torch.autograd.Function` whose backward runs the engine's explicit
  HIP backward and hands autograd views of the flat gradient buffer (recycled between steps unless a parameter's `.grad`
  still aliases it, so gradient accumulation over micro-batches is autograd's own `+=`);
* an optimizer writes the fp32 views in place; views share `P`'s version counter, so the next forward sees the bump and
  refreshes the bf16 MFMA operands (`engine.sync_weights()`).

The fused path (`engine.zero_grad / backward / optimizer_step`: one clip + AdamW launch over the flat buffers, gradient
exchange overlapped with the backward) stays available and is what `trainer.MyTrainer` and `bench.py` use.
"""
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn


class _Box(nn.Module):
    """container node of the parameter tree (e.g. `transformer.layers.3.dec_attn.qkv_net`)"""


class _EngineLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, run, *params):
        out = run()
        ctx.model = model
        ctx.token = model._fwd_token
        model._last_out = out
        return out['loss']

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        model = ctx.model
        if ctx.token != model._fwd_token:
            raise RuntimeError('backward() of a loss whose activations were overwritten by a later train-mode forward: the '
                               'engine keeps the activations of ONE forward (call backward before the next forward)')
        eng = model.engine
        # autograd may keep the returned views as the parameters' `.grad` (and then accumulates micro-batches itself with `+=`),
        # so the flat buffer can only be recycled when no parameter still holds a view of it -- the usual case, after
        # `optimizer.zero_grad()` (set_to_none) -- otherwise a fresh one is taken (372 MB at C3: not something to do per step)
        G = eng.G
        if G is not None and G.shape == eng.P.shape:
            base = G.untyped_storage().data_ptr()
            held = any(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in model.parameters())
        else:
            held = True
        if held:
            eng.G = torch.zeros_like(eng.P)
        else:
            G.zero_()
        sync = model._grad_sync
        scale = float(gout)
        if sync is not None:
            from . import dist as mdist
            scale /= mdist.world_size()          # mean over ranks: loss is a per-rank mean
        eng.backward(grad_scale=scale, layer_done=None if sync is None else sync.layer_done)
        if sync is not None:
            sync.finish()
        eng.rng_step += 1          # the next forward (micro-batch or step) draws fresh dropout masks / LSH rotations
        return (None, None) + tuple(eng.layout.view(eng.G, n) for n in model._param_names)


class EngineModule(nn.Module):
    """Base of `MyTransfoXLLMHeadModel` / `MyReformerModelWithLMHead`.  Subclasses set `self.engine` before calling
    `_bind_parameters()`; `tied` maps extra state-dict names onto an existing parameter (upstream's tied softmax weight)."""

    def __init__(self):
        super().__init__()
        self._fwd_token = 0
        self._last_out = None
        self._grad_sync = None
        self._synced_version = None
        self._param_names = []

    # ------------------------------------------------------------------ parameter tree
    def _bind_parameters(self, tied: Optional[dict] = None):
        eng = self.engine
        made = {}
        for name in eng.layout.real_names():
            p = nn.Parameter(eng.layout.view(eng.P, name), requires_grad=True)
            made[name] = p
            self._register_dotted(name, p)
        self._param_names = list(made)
        for alias, target in (tied or {}).items():
            self._register_dotted(alias, made[target])
        self._synced_version = eng.P._version

    def _register_dotted(self, name: str, p: nn.Parameter):
        node = self
        *path, leaf = name.split('.')
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Box())
            node = node._modules[part]
        node.register_parameter(leaf, p)

    def _flat_params(self):
        by_name = dict(self.named_parameters())
        return [by_name[n] for n in self._param_names]

    # ------------------------------------------------------------------ nn.Module plumbing
    def _apply(self, fn, recurse=True):
        """`.to(device)` / `.cuda()` on the engine's own device are no-ops; a dtype or device change would detach the
        parameters from the flat buffer the kernels read, so it is refused."""
        probe = fn(torch.empty(0, device=self.engine.P.device, dtype=self.engine.P.dtype))
        if probe.device != self.engine.P.device or probe.dtype != self.engine.P.dtype:
            raise RuntimeError('the HIP engine owns its parameters: fp32 masters on its GPU (bf16 operands are derived); '
                               'moving or casting the module is not supported')
        return self

    def _maybe_resync(self):
        eng = self.engine
        if eng.P._version != self._synced_version:     # an optimizer / load wrote the fp32 views in place
            eng.sync_weights()
            self._synced_version = eng.P._version

    def mark_synced(self):
        """the engine itself just refreshed the bf16 operands (fused optimizer step, load_state_dict)"""
        self._synced_version = self.engine.P._version

    def num_parameters(self, only_trainable: bool = False, exclude_embeddings: bool = False) -> int:
        return self.engine.num_parameters()

    def state_dict(self, *args, **kwargs):
        """CPU copies under upstream's names (what `save_pretrained` writes); nn.Module's own prefix / keep_vars forms go
        through the default implementation."""
        if args or kwargs:
            return super().state_dict(*args, **kwargs)
        sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in super().state_dict().items())
        return sd

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        self.engine.load_state_dict(sd, strict=strict)
        self.mark_synced()
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        self.engine.zero_grad()

    def backward(self, grad_scale: float = 1.0, layer_done=None):
        """fused path: gradients of the last train-mode forward accumulated into `engine.G`"""
        self.engine.backward(grad_scale=grad_scale, layer_done=layer_done)

    def enable_data_parallel(self):
        """`loss.backward()` then also averages the gradients over the ranks of the default process group (per-layer
        buckets overlapped with the rest of the backward, dist.GradSync) -- the role DDP plays for the reference's stack."""
        from .dist import GradSync
        self._grad_sync = GradSync(self.engine)
        return self

    # ------------------------------------------------------------------ autograd bridge
    def _run_engine(self, run, differentiable: bool):
        """`run()` -> the engine's output dict.  With grad enabled in train mode the loss is tied to the parameters through
        `_EngineLoss`."""
        self._maybe_resync()
        if differentiable and torch.is_grad_enabled():
            self._fwd_token += 1
            loss = _EngineLoss.apply(self, run, *self._flat_params())
            out = dict(self._last_out)
            out['loss'] = loss
            return out
        if differentiable:
            self._fwd_token += 1
        return run()
