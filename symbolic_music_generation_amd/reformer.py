"""Drop-in surface for `musicnlp.models.reformer` (reference: musicnlp/models/reformer.py:13-127): `MyReformerConfig`
(presets, derived fields, tokenizer ids, axial assert) and `MyReformerModelWithLMHead` (pass-through forward accepting
`key_scores`), on the HIP engine `rf_engine.RFEngine` instead of HuggingFace `ReformerModelWithLMHead`."""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch

from .module import EngineModule
from .rf_engine import RFEngine

__all__ = ['MyReformerConfig', 'MyReformerModelWithLMHead']


class MyReformerConfig:
    _layer_pair = ['local', 'lsh']
    presets = {   # reformer.py:15-44
        'debug': dict(max_position_embeddings=64, axial_pos_shape=(8, 8), hidden_size=128, num_attention_heads=8, attn_layers=_layer_pair * 3),
        'debug-large': dict(max_position_embeddings=512, axial_pos_shape=(16, 32), hidden_size=128, num_attention_heads=8, attn_layers=_layer_pair * 3),
        'tiny': dict(max_position_embeddings=1024, axial_pos_shape=(32, 32), hidden_size=256, num_attention_heads=8, attn_layers=_layer_pair * 3),
        'small': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=512, num_attention_heads=8, attn_layers=_layer_pair * 3),
        'base': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=768, num_attention_heads=12, attn_layers=_layer_pair * 6, num_hashes=2),
        'large': dict(max_position_embeddings=2048, axial_pos_shape=(32, 64), hidden_size=1024, num_attention_heads=16, attn_layers=_layer_pair * 12, num_hashes=2),
    }
    # HF ReformerConfig defaults in force (logged config, notebook/train/reformer.ipynb cell 10)
    _hf_defaults = dict(
        vocab_size=320, num_hashes=1, num_buckets=None, lsh_attn_chunk_length=64, local_attn_chunk_length=64,
        lsh_num_chunks_before=1, lsh_num_chunks_after=0, local_num_chunks_before=1, local_num_chunks_after=0,
        hidden_act='relu', hidden_dropout_prob=0.05, lsh_attention_probs_dropout_prob=0.0,
        local_attention_probs_dropout_prob=0.05, layer_norm_eps=1e-12, axial_norm_std=1.0, initializer_range=0.02,
        chunk_size_lm_head=0, chunk_size_feed_forward=0, tie_word_embeddings=False, hash_seed=None, axial_pos_embds=True,
        eos_token_id=2, pad_token_id=0, is_decoder=False,
    )
    model_type = 'reformer'

    def __init__(self, model_size: str = 'base', tokenizer=None, **kwargs):
        d_config = dict(self._hf_defaults)
        preset = {k: (list(v) if isinstance(v, list) else v) for k, v in MyReformerConfig.presets[model_size].items()}
        hd_sz, n_head = preset['hidden_size'], preset['num_attention_heads']
        assert hd_sz % n_head == 0 and hd_sz % 4 == 0                                  # :46-48
        preset.update(feed_forward_size=hd_sz * 4, attention_head_size=hd_sz // n_head,
                      axial_pos_embds_dim=(hd_sz // 4, 3 * hd_sz // 4), is_decoder=True, num_buckets=None)   # :49-55
        d_config.update(preset)
        if tokenizer is not None:                                                       # :59-67
            d_config.update(eos_token_id=tokenizer.eos_token_id, pad_token_id=tokenizer.pad_token_id,
                            vocab_size=tokenizer.vocab_size)
        d_config.update(kwargs)
        for k, v in d_config.items():
            setattr(self, k, v)
        self.model_size = model_size
        self.axial_pos_shape = tuple(self.axial_pos_shape)
        self.axial_pos_embds_dim = tuple(self.axial_pos_embds_dim)
        aps, mpe = self.axial_pos_shape, self.max_position_embeddings
        assert len(aps) == 2 and aps[0] * aps[1] == mpe, \
            'the product of `axial_pos_shape` must be `max_position_embeddings`'        # :71-73
        if self.hidden_act != 'relu' or self.chunk_size_lm_head or self.chunk_size_feed_forward:
            raise NotImplementedError('only the reference configuration (relu, no FF / LM-head chunking) is implemented')
        self.use_return_dict = True

    @property
    def max_length_(self) -> int:                                                       # :75-77
        return self.max_position_embeddings

    @property
    def model_meta(self) -> Dict[str, Any]:                                             # :79-87
        return dict(axial_pos_shape=self.axial_pos_shape, n_layer=len(self.attn_layers), hidden_size=self.hidden_size,
                    ff_size=self.feed_forward_size,
                    attention_shape=f'{self.num_attention_heads}x{self.attention_head_size}', vocab_size=self.vocab_size)

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith('_') and k != 'use_return_dict'}

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, 'config.json'), 'w') as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, 'config.json')) as f:
            d = json.load(f)
        return cls(model_size=d.pop('model_size', 'base'), **d)


@dataclass
class ReformerModelWithLMHeadOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_buckets_states: Any = None
    hidden_states: Any = None
    attentions: Any = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits) if v is not None)[k]


def c_max_len(cfg) -> int:
    return cfg.axial_pos_shape[0] * cfg.axial_pos_shape[1]


class MyReformerModelWithLMHead(EngineModule):
    """`torch.nn.Module` over `RFEngine` (see module.EngineModule): HF's state-dict names, autograd-connected loss."""
    cls_name = 'Reformer'

    def __init__(self, config: MyReformerConfig, device='cuda:0', seed: int = 77):
        super().__init__()
        self.config = config
        self.engine = RFEngine(config, device, seed=seed)
        self.device = torch.device(device)
        self._bind_parameters()

    def save_pretrained(self, path):
        self.config.save_pretrained(path)
        torch.save(self.state_dict(), os.path.join(path, 'pytorch_model.bin'))

    @classmethod
    def from_pretrained(cls, path, device='cuda:0'):
        m = cls(MyReformerConfig.from_pretrained(path), device=device)
        m.load_state_dict(torch.load(os.path.join(path, 'pytorch_model.bin'), map_location='cpu'))
        return m

    def forward(self, key_scores=None, input_ids=None, position_ids=None, attention_mask=None, head_mask=None,
                inputs_embeds=None, num_hashes=None, past_buckets_states=None, use_cache=None, output_hidden_states=None,
                output_attentions=None, return_dict=None, labels=None, rotations=None, buckets=None):
        """reformer.py:96-127.  The tokenizer never emits an attention mask (model_input_names = ['input_ids']), so pad
        tokens are attended to and only dropped from the loss -- as in the reference.  `rotations` / `buckets`
        ({lsh layer index: tensor}) make the hashing an explicit input for parity tests."""
        if input_ids is None:
            raise ValueError('input_ids required')
        if any(x is not None for x in (position_ids, attention_mask, head_mask, inputs_embeds, past_buckets_states)) \
                or use_cache or output_hidden_states or output_attentions:
            raise NotImplementedError('only the arguments the reference passes (input_ids, labels) are implemented')
        if num_hashes is not None and num_hashes != self.config.num_hashes:
            raise NotImplementedError('per-call num_hashes override')
        ids = input_ids.to(self.device)
        lab = None if labels is None else labels.to(self.device)
        out = self._run_engine(lambda: self.engine.forward(ids, labels=lab, train=self.training, rotations=rotations,
                                                           buckets_override=buckets),
                               differentiable=self.training and labels is not None)
        res = ReformerModelWithLMHeadOutput(loss=out['loss'], logits=out['logits'])
        return res[:] if return_dict is False else res

    @torch.no_grad()
    def generate(self, input_ids=None, max_length: Optional[int] = None, do_sample: bool = False, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, temperature: float = 1.0, repetition_penalty: Optional[float] = None,
                 typical_p: Optional[float] = None, seed: int = 77, use_cache: bool = True, rotations=None, **unsupported):
        """`model.generate(...)` as the reference drives it (musicnlp/trainer/eval.py:277-333): greedy, or sampling with
        top-k / top-p / typical-p / temperature / repetition penalty (applied, as HF does, to the raw logits); token selection
        runs on the device (the TransfoXL decoder's sampler kernel).

        use_cache=True (HF's default, what the reference gets): incremental decoding -- the prompt in one pass, then one token
        per step against per-layer caches of projections and LSH bucket ids (rf_generate.RFDecoder; HF `ReformerDynamicCache`).
        `rotations` ({lsh layer: (H, dh, n_h, rot/2)}) fixes the hash rotations (HF `config.hash_seed`); by default one seeded
        draw serves the whole generation.
        use_cache=False: every step is a full forward over the tokens so far, right-padded to a multiple of the chunk length
        (pads sit after every real token, so the causal mask keeps them out), with the rotations redrawn each forward."""
        from . import ops
        top_k = getattr(self.config, 'top_k', 50) if top_k is None else top_k        # HF fills it from the config: default 50
        if unsupported.get('penalty_alpha') and not do_sample and top_k is not None and top_k > 1:
            # HF 4.25.1 contrastive_search takes `past_buckets_states` as its cache and indexes past[0][0].shape: the bucket entry
            # of a local layer is None, so the reference stack fails there too; the ValueError is this package's
            raise ValueError(f"{type(self).__name__} **can't** be used for contrastive search: its cache (past_buckets_states) "
                             'is not a per-layer tensor cache that HF\'s routine can replicate per candidate')
        if unsupported.get('num_return_sequences', 1) not in (None, 1) and (unsupported.get('num_beams', 1) or 1) == 1:
            if not do_sample:
                raise ValueError('num_return_sequences has to be 1 when doing greedy search')
            input_ids = input_ids.repeat_interleave(int(unsupported.pop('num_return_sequences')), 0)
        num_beams = unsupported.pop('num_beams', 1) or 1
        num_beam_groups = unsupported.pop('num_beam_groups', 1) or 1
        diversity_penalty = unsupported.pop('diversity_penalty', None)
        if num_beam_groups != 1:
            # diverse (group) beam search (eval.py:303-317) over the cached decoder's beam hooks
            if num_beams <= 1 or num_beam_groups > num_beams:
                raise ValueError('`num_beam_groups` has to be smaller or equal to `num_beams`')
            if do_sample:
                raise ValueError('Diverse beam search cannot be used in sampling mode. Make sure that `do_sample` is set to `False`.')
            from .generate import group_beam_search
            from .rf_generate import RFDecoder
            nrs = int(unsupported.pop('num_return_sequences', 1) or 1)
            self._maybe_resync()
            was = self.training
            self.eval()
            try:
                max_length = int(max_length or c_max_len(self.config))
                dec = RFDecoder(self.engine, input_ids.shape[0] * num_beams, max_length, rotations=rotations, seed=seed)
                return group_beam_search(dec, input_ids, max_length, num_beams=num_beams, num_beam_groups=num_beam_groups,
                                         diversity_penalty=diversity_penalty or 0.0,
                                         early_stopping=bool(unsupported.get('early_stopping')),
                                         length_penalty=float(unsupported.get('length_penalty', 1.0) or 1.0),
                                         num_return_sequences=nrs, eos_token_id=self.config.eos_token_id,
                                         pad_token_id=self.config.pad_token_id)
            finally:
                if was:
                    self.train()
        if num_beams > 1:
            # the reference's 'beam' strategy (eval.py:302-321): HF beam_search / beam_sample over the cached decoder
            from .generate import beam_search
            from .rf_generate import RFDecoder
            nrs = int(unsupported.pop('num_return_sequences', 1) or 1)
            self._maybe_resync()
            was = self.training
            self.eval()
            try:
                max_length = int(max_length or c_max_len(self.config))
                dec = RFDecoder(self.engine, input_ids.shape[0] * num_beams * (nrs if do_sample else 1), max_length,
                                rotations=rotations, seed=seed)
                gen = torch.Generator(device=self.device).manual_seed(seed) if do_sample else None
                return beam_search(dec, input_ids, max_length, num_beams=num_beams, do_sample=do_sample, top_k=top_k, top_p=top_p,
                                   temperature=temperature, typical_p=typical_p,
                                   early_stopping=bool(unsupported.get('early_stopping')),
                                   length_penalty=float(unsupported.get('length_penalty', 1.0) or 1.0),
                                   renormalize_logits=bool(unsupported.get('renormalize_logits')), num_return_sequences=nrs,
                                   eos_token_id=self.config.eos_token_id, pad_token_id=self.config.pad_token_id, generator=gen)
            finally:
                if was:
                    self.train()
        if unsupported:
            ok = {'early_stopping', 'renormalize_logits'}      # no effect: eos never ends a row; the sampler always renormalises
            bad = [k for k, v in unsupported.items() if k not in ok and v not in (None, False, 1, 1.0)]
            if bad:
                raise NotImplementedError(f'generation options not covered: {bad}')
        self._maybe_resync()
        c = self.config
        was_training = self.training
        self.eval()
        ids0 = input_ids.to(self.device)
        B, Tp = ids0.shape
        A0, A1 = c.axial_pos_shape
        max_length = int(max_length or A0 * A1)
        if max_length > A0 * A1:
            raise ValueError('max_length exceeds max_position_embeddings')
        if max_length <= Tp:
            return ids0[:, :max_length]
        try:
            if use_cache:
                from .rf_generate import RFDecoder
                dec = getattr(self, '_decoder', None)
                if dec is None or dec.B != B or dec.Tmax < max_length:
                    dec = self._decoder = RFDecoder(self.engine, B, max_length, seed=seed)
                dec.rotations = rotations
                dec.seed = seed
                return dec.generate(ids0, max_length, do_sample=do_sample, top_k=top_k, top_p=top_p, temperature=temperature,
                                    repetition_penalty=repetition_penalty, typical_p=typical_p)
            V = c.vocab_size
            pad = getattr(c, 'pad_token_id', None)
            pad = 0 if pad is None else int(pad)
            buf = torch.full((B, max_length + 64), pad, device=self.device, dtype=torch.int64)
            buf[:, :Tp] = ids0
            t_dev = torch.full((1,), Tp - 1, device=self.device, dtype=torch.int32)
            rng = torch.zeros(1, device=self.device, dtype=torch.int64)
            for cur in range(Tp, max_length):
                Tf = cur if cur <= 64 else (cur + 63) // 64 * 64
                out = self.engine.forward(buf[:, :Tf].contiguous(), labels=None, train=False)
                last = out['logits'][:, cur - 1].contiguous()
                ops.sample(last, buf, t_dev, rng, seed, do_sample=do_sample, top_k=top_k or 0,
                           top_p=top_p if top_p is not None else 1.0, temperature=temperature,
                           repetition_penalty=repetition_penalty, typical_p=typical_p)
                ops.decode_advance(t_dev, rng)
                if Tf > cur:
                    buf[:, cur + 1:Tf] = pad          # keep the padding clean (the sampler wrote position `cur` only)
            return buf[:, :max_length].clone()
        finally:
            if was_training:
                self.train()
