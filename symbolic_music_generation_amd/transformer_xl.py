"""Drop-in surface for `musicnlp.models.transformer_xl` (reference: musicnlp/models/transformer_xl.py:15-241).

Same names, constructor arguments, presets and output fields as the reference's `MyTransfoXLConfig` /
`MyTransfoXLLMHeadModel`; the arithmetic the reference inherits from HuggingFace `TransfoXLLMHeadModel` runs here on
the HIP engine (`xl_engine.XLEngine`).  Public tensors keep the reference's conventions: `input_ids`/`labels` (B, T)
int64, `mems` = list of n_layer tensors shaped (mem_len, B, d_model) (time-major, as upstream returns them),
`prediction_scores` = log-probabilities (B, T, V).
"""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from .module import EngineModule
from .xl_engine import XLEngine

__all__ = ['MyTransfoXLConfig', 'MyTransfoXLLMHeadModel', 'TransfoXLLMHeadModelOutput']


class MyTransfoXLConfig:
    # reference presets, transformer_xl.py:16-23
    presets = {
        'debug': dict(d_model=128, n_head=8, n_layer=4),
        'debug-large': dict(d_model=128, n_head=8, n_layer=4),
        'tiny': dict(d_model=256, n_head=8, n_layer=6),
        'small': dict(d_model=512, n_head=8, n_layer=12),
        'base': dict(d_model=768, n_head=12, n_layer=12),
        'large': dict(d_model=1024, n_head=16, n_layer=18),
    }
    size2max_length = {'debug': 64, 'debug-large': 128, 'tiny': 512, 'small': 1024, 'base': 2048, 'large': 2048}
    model_type = 'transfo-xl'

    # upstream TransfoXLConfig defaults that stay in force (logged config, notebook/train/transformer-xl.ipynb cell 10)
    _hf_defaults = dict(
        vocab_size=267735, cutoffs=[20000, 40000, 200000], same_length=True, attn_type=0, untie_r=True, pre_lnorm=False,
        dropout=0.1, dropatt=0.0, adaptive=True, sample_softmax=-1, tie_projs=[False], tie_word_embeddings=True,
        layer_norm_epsilon=1e-5, init='normal', init_range=0.01, init_std=0.02, proj_init_std=0.01, eos_token_id=0,
        pad_token_id=None, div_val=1, proj_share_all_but_first=True,
    )

    def __init__(self, model_size: str = 'base', tokenizer=None, max_length: int = None, **kwargs):
        config = dict(self._hf_defaults)
        preset = dict(MyTransfoXLConfig.presets[model_size])
        hd_sz, n_head = preset['d_model'], preset['n_head']
        assert hd_sz % n_head == 0
        if 'debug' in model_size:
            m_len, c_len = 64, 64                                                  # :30-31
        else:
            m_len = max(128, self.size2max_length[model_size] // 8)               # :33
            c_len = max(1024, self.size2max_length[model_size] // 2)              # :34
        preset.update(d_embed=hd_sz, d_inner=hd_sz * 4, d_head=hd_sz // n_head, mem_len=m_len, clamp_len=c_len, div_val=1)
        config.update(preset)
        if tokenizer is not None:                                                  # :55-66
            vsz = config['vocab_size'] = tokenizer.vocab_size
            if vsz >= 32768 * 8:
                config['cutoffs'] = [20000, 40000, 200000]
            elif vsz >= 32768:
                config['cutoffs'] = [10000]
            elif vsz >= 16384:
                config['cutoffs'] = [5000]
            elif vsz >= 1000:
                config['cutoffs'] = [1000]
            else:
                config['cutoffs'] = []
        config.update(kwargs)                                                      # :67
        for k, v in config.items():
            setattr(self, k, v)
        self.model_size = model_size
        self.max_length_ = max_length or MyTransfoXLConfig.size2max_length[model_size]  # :70
        self.use_return_dict = True
        if self.dropatt != 0.0:
            raise NotImplementedError('dropatt != 0 is never used by the reference (HF default 0.0)')
        if len(self.cutoffs) > 3:
            raise NotImplementedError('at most 3 adaptive-softmax cutoffs')

    @property
    def model_meta(self) -> Dict[str, Any]:                                        # :72-77
        return dict(n_layer=self.n_layer, hidden_size=self.d_embed, ff_size=self.d_inner, seg_len=self.mem_len,
                    max_len=self.max_length_, vocab_size=self.vocab_size)

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if not k.startswith('_') and k != 'use_return_dict'}

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, 'config.json'), 'w') as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def from_pretrained(cls, path: str) -> 'MyTransfoXLConfig':
        with open(os.path.join(path, 'config.json')) as f:
            d = json.load(f)
        size = d.pop('model_size', 'base')
        max_length = d.pop('max_length_', None)
        return cls(model_size=size, max_length=max_length, **d)


@dataclass
class TransfoXLLMHeadModelOutput:
    """Fields of the reference's output class (transformer_xl.py:81-124)."""
    losses: Optional[torch.Tensor] = None
    prediction_scores: Any = None
    mems: Optional[List[torch.Tensor]] = None
    hidden_states: Any = None
    attentions: Any = None
    loss: Optional[torch.Tensor] = None

    @property
    def logits(self):  # log-probabilities, "behave the same way logits do" (:117-124)
        return self.prediction_scores

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.prediction_scores, self.losses, self.mems) if v is not None)[k]


class MyTransfoXLLMHeadModel(EngineModule):
    """`torch.nn.Module` (module.EngineModule): parameters are fp32 views into the engine's flat buffer under upstream's
    state-dict names, and a train-mode `loss` carries an autograd node whose backward is the engine's HIP backward -- the
    reference's `Trainer` loop (`loss.backward(); clip_grad_norm_; optimizer.step()`) drives it unchanged."""
    cls_name = 'TransformerXl'

    def __init__(self, config: MyTransfoXLConfig, device='cuda:0', seed: int = 77):
        super().__init__()
        self.config = config
        self.engine = XLEngine(config, device, seed=seed)
        self.device = torch.device(device)
        # upstream ties crit.out_layers.0.weight to the embedding (tie_word_embeddings, div_val == 1)
        self._bind_parameters(tied={'crit.out_layers.0.weight': 'transformer.word_emb.emb_layers.0.weight'})

    def save_pretrained(self, path: str):
        """HF layout: config.json + pytorch_model.bin with upstream parameter names."""
        self.config.save_pretrained(path)
        torch.save(self.state_dict(), os.path.join(path, 'pytorch_model.bin'))

    @classmethod
    def from_pretrained(cls, path: str, device='cuda:0'):
        config = MyTransfoXLConfig.from_pretrained(path)
        model = cls(config, device=device)
        model.load_state_dict(torch.load(os.path.join(path, 'pytorch_model.bin'), map_location='cpu'))
        return model

    # -- mems: API is upstream's time-major list; the engine is batch-major
    @staticmethod
    def _mems_in(mems):
        return None if mems is None else [m.transpose(0, 1).contiguous().to(torch.bfloat16) for m in mems]

    @staticmethod
    def _mems_out(mems):
        return None if mems is None else [m.transpose(0, 1) for m in mems]

    def forward(self, key_scores=None, input_ids: Optional[torch.Tensor] = None, mems=None, head_mask=None,
                inputs_embeds=None, labels: Optional[torch.Tensor] = None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        """Same contract as the reference forward (transformer_xl.py:130-221)."""
        if input_ids is None:
            raise ValueError('You have to specify input_ids (inputs_embeds is not used by the reference call sites)')
        if head_mask is not None or inputs_embeds is not None or output_attentions or output_hidden_states:
            raise NotImplementedError('head_mask / inputs_embeds / attention & hidden-state outputs are never requested '
                                      'by the reference (ignore_keys_for_eval, train.py:588)')
        input_ids = input_ids.to(self.device)
        if labels is not None:
            labels = labels.to(self.device)
        if mems is not None and len(mems) and mems[0].size(0) != self.config.mem_len:
            raise ValueError('mems must hold exactly mem_len rows (upstream init_mems/_update_mems invariant)')
        mems_in = self._mems_in(mems)
        out = self._run_engine(lambda: self.engine.forward(input_ids, mems=mems_in, labels=labels, train=self.training),
                               differentiable=self.training and labels is not None)
        in_eval = not self.training
        prediction_scores = out['logprobs'] if (labels is None or in_eval) else ()
        res = TransfoXLLMHeadModelOutput(loss=out['loss'], prediction_scores=prediction_scores, losses=out['losses'],
                                         mems=self._mems_out(out['mems']))
        if return_dict is False:
            return res[:]
        return res

    def prepare_inputs_for_generation(self, input_ids, past=None, **model_kwargs):
        """transformer_xl.py:223-241"""
        inputs = {}
        if past:
            assert isinstance(past, list)
            if isinstance(past[0], list):
                past = [torch.stack(p, dim=0) for p in past]
            inputs['mems'] = past
            inputs['input_ids'] = input_ids[:, -1].unsqueeze(-1)
        else:
            inputs['input_ids'] = input_ids
        return inputs

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor = None, max_length: int = None, do_sample: bool = False,
                 top_k: Optional[int] = None, top_p: Optional[float] = None, temperature: float = 1.0, num_beams: int = 1,
                 penalty_alpha=None, typical_p=None, repetition_penalty=None, early_stopping=None,
                 renormalize_logits=None, num_return_sequences: int = 1, num_beam_groups: int = 1, length_penalty: float = 1.0,
                 use_graph: bool = True, seed: int = 77, **unused) -> torch.Tensor:
        """`model.generate(**inputs, **args)` as called at musicnlp/trainer/eval.py:333: the greedy, sample, contrastive and beam
        strategies (eval.py:277-321), beam search in its plain, sampling and diverse-group forms.  `num_return_sequences` expands the
        prompts as HF does (repeat_interleave).  Mode selection follows HF 4.25.1 `generate`: contrastive search when
        `penalty_alpha > 0`, `top_k > 1`, `do_sample` false and one beam; group beam search when `num_beam_groups > 1`."""
        from .generate import XLDecoder, XLDecoderLanes, beam_search, contrastive_search, group_beam_search
        # HF 4.25.1 fills unspecified generation arguments from the model config; PretrainedConfig's default top_k is 50, so
        # `generate(do_sample=True)` without top_k samples from the 50 best tokens (the reference relies on these defaults)
        top_k = getattr(self.config, 'top_k', 50) if top_k is None else top_k
        diversity_penalty = unused.pop('diversity_penalty', None)
        self._maybe_resync()
        max_length = max_length or self.config.max_length_
        if penalty_alpha is not None and penalty_alpha > 0 and top_k is not None and top_k > 1 and not do_sample and num_beams == 1:
            dec = XLDecoder(self.engine, input_ids.shape[0] * top_k, max_length, seed=seed)
            return contrastive_search(dec, input_ids, max_length, top_k=top_k, penalty_alpha=penalty_alpha,
                                      eos_token_id=self.config.eos_token_id, pad_token_id=self.config.pad_token_id)
        if num_beam_groups != 1:
            if num_beams <= 1 or num_beam_groups > num_beams:
                raise ValueError('`num_beam_groups` has to be smaller or equal to `num_beams`')               # HF's message
            if do_sample:
                raise ValueError('Diverse beam search cannot be used in sampling mode. Make sure that `do_sample` is set to `False`.')
            dec = XLDecoder(self.engine, input_ids.shape[0] * num_beams, max_length, seed=seed)
            return group_beam_search(dec, input_ids, max_length, num_beams=num_beams, num_beam_groups=num_beam_groups,
                                     diversity_penalty=diversity_penalty or 0.0, early_stopping=bool(early_stopping),
                                     length_penalty=length_penalty, num_return_sequences=num_return_sequences,
                                     eos_token_id=self.config.eos_token_id, pad_token_id=self.config.pad_token_id)
        if num_beams > 1:
            rows = input_ids.shape[0] * num_beams * (num_return_sequences if do_sample else 1)
            dec = XLDecoder(self.engine, rows, max_length, seed=seed)
            gen = torch.Generator(device=self.device).manual_seed(seed) if do_sample else None
            return beam_search(dec, input_ids, max_length, num_beams=num_beams, do_sample=do_sample, top_k=top_k, top_p=top_p,
                               temperature=temperature, typical_p=typical_p, early_stopping=bool(early_stopping),
                               renormalize_logits=bool(renormalize_logits),
                               length_penalty=length_penalty, num_return_sequences=num_return_sequences,
                               eos_token_id=self.config.eos_token_id, pad_token_id=self.config.pad_token_id, generator=gen)
        if num_return_sequences > 1:
            if not do_sample:
                raise ValueError('num_return_sequences has to be 1 when doing greedy search')       # HF's message
            input_ids = input_ids.repeat_interleave(num_return_sequences, 0)
        B = input_ids.shape[0]
        dec = getattr(self, '_decoder', None)
        # two free-running half-batch lanes from 32 rows on (generate.XLDecoderLanes); MXL_DECODE_LANES=1 keeps one decoder
        lanes = max(1, int(os.environ.get('MXL_DECODE_LANES', '2'))) if (B >= 32 and use_graph) else 1
        if dec is None or dec.B != B or dec.Tmax < max_length or getattr(dec, 'n', 1) != lanes:
            dec = self._decoder = (XLDecoderLanes(self.engine, B, max_length, seed=seed, lanes=lanes) if lanes > 1
                                   else XLDecoder(self.engine, B, max_length, seed=seed))
        dec.invalidate_tables()
        return dec.generate(input_ids.to(self.device), max_length, do_sample=do_sample, top_k=top_k, top_p=top_p,
                            temperature=temperature, repetition_penalty=repetition_penalty, typical_p=typical_p,
                            use_graph=use_graph)
