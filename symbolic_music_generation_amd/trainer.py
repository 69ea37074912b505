"""Drop-in training surface for `musicnlp.trainer.train` (reference: musicnlp/trainer/train.py:31-368 and the
`MyTrainer` of musicnlp/util/train/train_util_wrap.py:41-144).

The reference drives HuggingFace `Trainer`; its optimisation semantics (AdamW beta 0.9/0.999 eps 1e-8, clip 1.0, cosine
schedule with warm-up ratio or constant for the debug presets, per-size batch/lr/weight-decay presets, epoch eval + save,
seed 77) are restated here around the HIP engine with explicit RCCL data parallelism.  No `Trainer`, no autograd.
"""
import json
import math
import os
import time
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch

from . import dist as mdist
from .transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
from .vocab import MusicTokenizer

PT_LOSS_PAD = -100  # musicnlp/util/train/train_util_wrap.py:22
RANDOM_SEED = 77    # musicnlp/util/config.json "random-seed"


def get_model_n_tokenizer(model_name: str, model_size: str, prec: int = 5, tokenize_scheme: str = 'vanilla',
                          tokenizer_filename: Optional[str] = None, pitch_kind: str = None,
                          tempo_bin: Union[bool, int] = None, model_config: Dict = None, device='cuda:0'):
    """train.py:31-59.  Returns (tokenizer, model, model_meta)."""
    if model_name not in ('transf-xl', 'reformer'):
        raise ValueError(f'Model Name mismatch: {model_name!r} not in [transf-xl, reformer]')
    if tokenize_scheme not in ('vanilla', 'wordpiece', 'pairmerge'):
        raise ValueError(f'Tokenization Scheme mismatch: {tokenize_scheme!r} not in [vanilla, wordpiece, pairmerge]')
    if tokenize_scheme == 'vanilla':
        tokenizer = MusicTokenizer(precision=prec, pitch_kind=pitch_kind or 'midi', tempo_bin=tempo_bin)
    else:       # train.py:38-45: a trained sub-word tokenizer loaded from its file (the reference ships none: train one with
        # subword.PairMergeTokenizerTrainer / WordPieceMusicTokenizerTrainer)
        from .subword import PairMergeTokenizer, WordPieceMusicTokenizer
        if not tokenizer_filename:
            raise ValueError(f'tokenize_scheme={tokenize_scheme!r} needs `tokenizer_filename`: the path of a trained tokenizer')
        cls = WordPieceMusicTokenizer if tokenize_scheme == 'wordpiece' else PairMergeTokenizer
        kw = dict(tempo_bin=tempo_bin, **(dict(pitch_kind=pitch_kind) if pitch_kind else {}))
        tokenizer = cls.from_file(tokenizer_filename, **kw)
    assert tokenizer.precision == prec
    if model_name == 'transf-xl':
        cls_config, cls_model = MyTransfoXLConfig, MyTransfoXLLMHeadModel
    else:
        from .reformer import MyReformerConfig, MyReformerModelWithLMHead
        cls_config, cls_model = MyReformerConfig, MyReformerModelWithLMHead
    config = cls_config(model_size=model_size, tokenizer=tokenizer, **(model_config or dict()))
    tokenizer.model_max_length = max_length = config.max_length_
    model_meta = OrderedDict({'model name': cls_model.cls_name, 'max length': max_length})
    model_meta.update(config.model_meta)
    return tokenizer, cls_model(config=config, device=device), model_meta


class TrainArgs:
    """train.py:62-228: per-size presets on top of the shared defaults."""
    _big = dict(batch_size=32, learning_rate=3e-4, weight_decay=1e-2, lr_scheduler_type='cosine', num_train_epochs=64,
                warmup_ratio=0.1)
    model_name2preset = {
        'transf-xl': {
            'debug': dict(batch_size=2, learning_rate=1e-3, weight_decay=0, lr_scheduler_type='constant', num_train_epochs=64),
            'debug-large': dict(batch_size=8, learning_rate=1e-3, weight_decay=0, lr_scheduler_type='constant', num_train_epochs=16),
            'tiny': dict(_big), 'small': dict(_big), 'base': dict(_big), 'large': dict(_big),
        },
        'reformer': {
            'debug': dict(batch_size=8, learning_rate=1e-3, weight_decay=0, lr_scheduler_type='constant', num_train_epochs=32),
            'debug-large': dict(batch_size=8, learning_rate=1e-3, weight_decay=0, lr_scheduler_type='constant', num_train_epochs=32),
            'tiny': dict(_big, num_train_epochs=32), 'small': dict(_big), 'base': dict(_big), 'large': dict(_big),
        },
    }

    def __init__(self, model_name: str, model_size: str):
        self.model_name, self.model_size = model_name, model_size

    @staticmethod
    def _get_default(model_name: str) -> Dict:  # train.py:165-190
        return dict(do_train=True, do_eval=True, evaluation_strategy='epoch', adam_beta1=0.9, adam_beta2=0.999,
                    adam_epsilon=1e-8, max_grad_norm=1, warmup_ratio=1e-2, logging_strategy='steps', logging_steps=1,
                    save_strategy='epoch', bf16=True, gradient_accumulation_steps=1, load_best_model_at_end=True,
                    metric_for_best_model='eval_loss', greater_is_better=False, output_dir=None)

    def __call__(self, train_args: Dict = None, n_train: int = None, my_train_args: Dict = None):
        """train.py:192-229.  With `my_train_args` given returns (args, my_args) like the reference's
        `TrainArgs.__call__(train_args, my_train_args, train_dataset)`; without, the args dict alone."""
        args = self._get_default(self.model_name)
        preset = dict(TrainArgs.model_name2preset[self.model_name][self.model_size])
        if 'batch_size' in preset:
            bsz = preset.pop('batch_size')
            preset['per_device_train_batch_size'] = preset['per_device_eval_batch_size'] = bsz
        args.update(preset)
        if train_args:
            args.update(train_args)
        steps_per_epoch = None
        if n_train is not None:
            bsz = args['per_device_train_batch_size'] * args.get('gradient_accumulation_steps', 1) * mdist.world_size()
            args['steps_per_epoch'] = steps_per_epoch = math.ceil(n_train / bsz)
        if my_train_args is None:
            return args
        my_args = dict(logging_strategy='steps', tqdm=False, insert_key=False, proportional_mixing=False)      # :208-211
        my_args.update(my_train_args)
        my_args['steps_per_epoch'] = steps_per_epoch
        save_epochs = my_args.get('save_epochs')
        if save_epochs:                                                                                         # :214-221
            assert args.get('save_strategy') == 'epoch', 'save per k epochs: save_strategy must be "epoch"'
            if save_epochs > 1 and steps_per_epoch:
                args['save_strategy'], args['save_steps'] = 'steps', save_epochs * steps_per_epoch
        if my_args['logging_strategy'] not in ('steps', 'epoch', 'no'):
            raise ValueError(f'logging_strategy {my_args["logging_strategy"]!r}')
        if my_args['logging_strategy'] == 'epoch' and steps_per_epoch:
            my_args['logging_steps'] = steps_per_epoch
        return {k: v for k, v in args.items() if v is not None}, my_args


def get_train_and_my_train_args(model_name: str, model_size: str, train_args: Dict = None, my_train_args: Dict = None,
                                train_dataset=None):
    """train.py:232-246"""
    n = len(train_dataset) if train_dataset is not None else None
    if n is not None and hasattr(train_dataset, 'n_rows'):
        n = train_dataset.n_rows()
    return TrainArgs(model_name, model_size)(train_args, n_train=n, my_train_args=my_train_args or dict())


def lr_at(step: int, total_steps: int, base_lr: float, scheduler: str, warmup_ratio: float) -> float:
    """HF get_scheduler('cosine' | 'constant' | 'linear') with warm-up = ceil(total * ratio); `step` counts finished steps."""
    warm = math.ceil(total_steps * warmup_ratio)
    if scheduler == 'constant':
        return base_lr
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    if scheduler == 'cosine':
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    if scheduler == 'linear':
        return base_lr * max(0.0, 1.0 - prog)
    raise ValueError(scheduler)


def collate_clm(batch_ids: torch.Tensor, pad_token_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """DataCollatorForLanguageModeling(mlm=False) as configured at train.py:360: labels = input_ids, pad -> -100."""
    labels = batch_ids.clone()
    labels[labels == pad_token_id] = PT_LOSS_PAD
    return batch_ids, labels


def ntp_accuracy(pred_ids: torch.Tensor, labels: torch.Tensor) -> float:
    """next-token accuracy over non-pad labels (train_util_wrap.py:115-121 / train.py:277-283)."""
    preds, lab = pred_ids[:, :-1], labels[:, 1:]
    msk = lab != PT_LOSS_PAD
    return (preds[msk] == lab[msk]).float().mean().item() if msk.any() else float('nan')


class MyTrainer:
    """One optimisation step = fwd -> loss -> bwd -> (RCCL all-reduce overlapped with bwd) -> clip -> AdamW -> lr step.

    `train_dataset` / `eval_dataset`: any sequence of equal-length int64 id tensors (already padded to max_length, the
    contract of musicnlp/preprocess/dataset.py:361).  Under data parallelism each rank takes a strided shard.
    """

    def __init__(self, model, tokenizer: Optional[MusicTokenizer], train_dataset=None, eval_dataset=None,
                 train_args: Dict = None, model_name: str = 'transf-xl', model_size: str = 'base',
                 log_fn: Optional[Callable[[Dict], None]] = None, seed: int = RANDOM_SEED, my_args: Dict = None,
                 model_meta: Dict = None):
        """`train_dataset` / `eval_dataset`: a sequence of equal-length id tensors, or a `data.DeviceBatcher` (token file ->
        augmentation -> device batches; it shards by rank itself)."""
        self.model, self.tokenizer = model, tokenizer
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.my_args, self.model_meta = dict(my_args or {}), model_meta
        n_train = None
        if train_dataset is not None:
            n_train = train_dataset.n_rows() if hasattr(train_dataset, 'n_rows') else len(train_dataset)
        self.args = TrainArgs(model_name, model_size)(train_args, n_train=n_train)
        self.engine = model.engine
        # gradient exchange: fp32 as HF's DDP unless `grad_exchange_dtype='bf16'` is asked for (half the bytes, ~2^-9 relative
        # rounding per rank's contribution: tests/test_dp_cpu.py::test_bf16_exchange_error_bound)
        self.sync = mdist.GradSync(self.engine, dtype=self.args.get('grad_exchange_dtype'))
        self.log_fn = log_fn or (lambda d: None)
        self.seed = seed
        self.global_step = 0
        self.log_history: List[Dict] = []
        self.pad_id = tokenizer.pad_token_id if tokenizer is not None else -1
        self.best = (float('inf'), None)

    # ---------------------------------------------------------------- one step
    def training_step(self, input_ids: torch.Tensor, labels: torch.Tensor, lr: float, micro: int = 0, n_micro: int = 1) -> torch.Tensor:
        """One micro-batch.  With `gradient_accumulation_steps` = k (HF semantics) the gradients of k consecutive micro-batches are
        summed in the flat buffer (each backward scaled 1/k), the ranks exchange once -- behind the LAST micro-batch's backward --
        and clip + AdamW run once; `rng_step` advances with every backward so that micro-batches draw their own dropout masks."""
        a = self.args
        self.model.train()
        if micro == 0:
            self.engine.zero_grad()
        last = micro == n_micro - 1
        out = self.model(input_ids=input_ids, labels=labels)
        self.engine.backward(grad_scale=1.0 / n_micro, layer_done=self.sync.layer_done if last else None)
        if not last:
            self.engine.rng_step += 1
            return out.loss
        self.sync.finish()
        self.engine.optimizer_step(lr=lr, betas=(a['adam_beta1'], a['adam_beta2']), eps=a['adam_epsilon'],
                                   weight_decay=a['weight_decay'], max_grad_norm=a['max_grad_norm'],
                                   grad_scale=1.0 / mdist.world_size())
        self.global_step += 1
        return out.loss

    def _shard(self, n: int, epoch: int, shuffle: bool, pad: bool) -> List[int]:
        """This rank's dataset indices.  Training (`pad=True`) follows `DistributedSampler`: the permuted index is padded by
        wrapping around to a multiple of the world size before striding, so EVERY rank gets ceil(n / world) samples and hence
        the same number of (equally sized) batches -- the per-layer gradient all-reduces of the ranks always pair up.
        Evaluation takes the plain strided shard (no duplicated samples; its only collective is one sum at the end)."""
        g = torch.Generator().manual_seed(self.seed + epoch)
        idx = torch.randperm(n, generator=g).tolist() if shuffle else list(range(n))
        world = mdist.world_size()
        if pad and n % world:
            idx += idx[:world - n % world]
        return idx[mdist.rank()::world]

    def _batches(self, ds, bsz: int, epoch: int, shuffle: bool, pad: bool = False, with_index: bool = False):
        """HF Trainer's loader keeps the last partial batch (`dataloader_drop_last=False`)."""
        if hasattr(ds, 'n_rows'):            # a DeviceBatcher: already sharded, shuffled, augmented, collated, on the device
            # ProportionalMixCallback.on_epoch_begin (train_util_wrap.py:307-317) re-draws the TRAIN sub-sample every epoch; the
            # eval dataset keeps the draw of its construction, so eval_loss compares the same subset from epoch to epoch
            if shuffle and hasattr(ds.tf, 'sample'):
                ds.tf.sample(epoch)
            ds.epoch = epoch
            for ids, labels in ds:
                yield ((ids, labels), None) if with_index else (ids, labels)
            return
        idx = self._shard(len(ds), epoch, shuffle, pad)
        for i in range(0, len(idx), bsz):
            rows = [torch.as_tensor(ds[j]) for j in idx[i:i + bsz]]
            if rows:
                b = torch.stack(rows).to(self.model.device)
                yield (b, idx[i:i + bsz]) if with_index else b

    # ---------------------------------------------------------------- loops
    def train(self, max_steps: Optional[int] = None) -> Dict:
        a = self.args
        bsz = a['per_device_train_batch_size']
        gas = max(1, int(a.get('gradient_accumulation_steps', 1)))
        n_train = self.train_dataset.n_rows() if hasattr(self.train_dataset, 'n_rows') else len(self.train_dataset)
        # optimizer steps per epoch == TrainArgs.steps_per_epoch (train.py:196-199): HF counts micro-batches // accumulation
        n_micro_epoch = max(1, math.ceil(n_train / (bsz * mdist.world_size())))
        spe = max(1, math.ceil(n_train / (bsz * gas * mdist.world_size())))
        total = max_steps or spe * int(a['num_train_epochs'])
        save_steps = int(a['save_steps']) if a.get('save_strategy') == 'steps' and a.get('save_steps') else None
        t0 = time.time()
        done = False
        for epoch in range(int(a['num_train_epochs'])):
            micro, seen = 0, 0
            for ids in self._batches(self.train_dataset, bsz, epoch, shuffle=True, pad=True):
                ids, labels = ids if isinstance(ids, tuple) else collate_clm(ids, self.pad_id)
                seen += 1
                # the last group of an epoch may hold fewer than `gas` micro-batches (HF steps on the epoch's last batch)
                group = min(gas, n_micro_epoch - (seen - 1 - micro))
                lr = lr_at(self.global_step, total, a['learning_rate'], a['lr_scheduler_type'], a['warmup_ratio'])
                step_before = self.global_step
                loss = self.training_step(ids, labels, lr, micro=micro, n_micro=group)
                micro = 0 if self.global_step != step_before else micro + 1
                if self.global_step == step_before:
                    continue
                if self.my_args.get('logging_strategy', 'steps') != 'no' and \
                        self.global_step % self.my_args.get('logging_steps', a['logging_steps']) == 0:
                    d = dict(step=self.global_step, epoch=epoch + self.global_step / spe % 1, learning_rate=lr,
                             loss=loss.item())
                    self.log_history.append(d)
                    self.log_fn(d)
                if save_steps and a['output_dir'] and self.global_step % save_steps == 0 and mdist.rank() == 0:
                    self.save_model(os.path.join(a['output_dir'], f'checkpoint-{self.global_step}'))
                if self.global_step >= total:
                    done = True
                    break
            if a['do_eval'] and self.eval_dataset is not None:
                ev = self.evaluate()
                ev.update(step=self.global_step, epoch=epoch + 1)
                self.log_history.append(ev)
                self.log_fn(ev)
                # checkpoint cadence: every epoch (save_strategy='epoch'), or every `save_steps` optimizer steps when
                # TrainArgs turned `save_epochs` = k into a step count (train.py:213-220: k epochs = k * steps_per_epoch)
                saves_now = a.get('save_strategy') == 'epoch' or (save_steps and self.global_step % save_steps == 0)
                if a['output_dir'] and saves_now:
                    # eval_loss is all-reduced and the path is deterministic: every rank tracks the same best checkpoint
                    ck = os.path.join(a['output_dir'], f'checkpoint-{self.global_step}')
                    if mdist.rank() == 0:
                        self.save_model(ck)
                    if ev['eval_loss'] < self.best[0]:
                        self.best = (ev['eval_loss'], ck)
            if done:
                break
        if a['load_best_model_at_end'] and self.best[1] is not None:
            if mdist.is_dist():
                torch.distributed.barrier()         # rank 0 has finished writing before anyone reads
            self.model.load_state_dict(torch.load(os.path.join(self.best[1], 'pytorch_model.bin'), map_location='cpu'))
        return dict(train_runtime=time.time() - t0, global_step=self.global_step)

    @torch.no_grad()
    def evaluate(self, key_scores=None) -> Dict:
        """eval loss + next-token accuracy (+ in-key ratio when `key_scores` (N_eval, 24) is given, the 'vanilla' IKR mode):
        argmax and the per-token counting run on the device (metrics.py), only (B, 14) integers per batch reach the host and
        only sums cross ranks (the reference gathers the full (B, T, V) logits: trainer_eval_wrap.py:310-314)."""
        from .metrics import ComputeMetrics, max_out_logits
        self.model.eval()
        bsz = self.args['per_device_eval_batch_size']
        cm = ComputeMetrics(self.tokenizer, mode='vanilla', clm_pred_shifted=False) if self.tokenizer is not None else None
        tot_loss, tot_n, hit, cnt, ikr_sum, ikr_n = 0.0, 0, 0.0, 0.0, 0.0, 0
        for ids, rows in self._batches(self.eval_dataset, bsz, 0, shuffle=False, with_index=True):
            ids, labels = ids if isinstance(ids, tuple) else collate_clm(ids, self.pad_id)
            out = self.model(input_ids=ids, labels=labels)
            preds = max_out_logits(out.logits)
            if cm is not None:
                c = cm.counts(preds, labels).cpu().numpy()
                hit += float(c[:, 12].sum()); cnt += float(c[:, 13].sum())
                if key_scores is not None and rows is not None:
                    ks = key_scores[rows]            # the shard is strided: index by the samples' dataset rows
                    ikr_sum += cm.ikr_from_counts(c, labels, ks) * ids.shape[0]
                    ikr_n += ids.shape[0]
            else:       # no tokenizer: accuracy only, still on device ids
                msk = labels[:, 1:] != PT_LOSS_PAD
                hit += (preds[:, :-1][msk] == labels[:, 1:][msk]).float().sum().item()
                cnt += msk.sum().item()
            tot_loss += out.loss.item() * ids.shape[0]
            tot_n += ids.shape[0]
        stats = torch.tensor([tot_loss, tot_n, hit, cnt, ikr_sum, ikr_n], dtype=torch.float64, device=self.model.device)
        if mdist.is_dist():
            torch.distributed.all_reduce(stats)
        s = stats.tolist()
        self.model.train()
        res = dict(eval_loss=s[0] / max(s[1], 1), eval_ntp_acc=s[2] / max(s[3], 1))
        if s[5] > 0:
            res['eval_ikr'] = s[4] / s[5]
        return res

    def save_model(self, path: str):
        self.model.save_pretrained(path)
        with open(os.path.join(path, 'trainer_state.json'), 'w') as f:
            json.dump(dict(global_step=self.global_step, log_history=self.log_history[-50:]), f)


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's entry points (musicnlp/trainer/train.py:287-368, 417-593).  The reference loads its HuggingFace datasets by
# name from its own processed-corpus directory (music21 extraction output: out of scope); here `dataset_names` names
# pre-tokenised token files (data.write_token_file) in the stored vocabulary -- 'step' pitches whenever pitch shift is on,
# as in the reference (dataset.py:259-262) -- and `dataset_args['keys']` carries the per-song key annotation that KeyInsert /
# PitchShift read from the reference's `keys` column.
# ---------------------------------------------------------------------------------------------------------------------------
def _open_split(names, split: str):
    from .data import TokenFile
    if isinstance(names, dict):
        names = names[split]
    if isinstance(names, (str, os.PathLike)) or hasattr(names, 'offsets'):
        names = [names]
    files = []
    for nm in names:
        if hasattr(nm, 'offsets') or hasattr(nm, 'files'):
            files.append(nm)
        else:
            path = str(nm)
            cand = [path, os.path.join(path, split), f'{path}.{split}', f'{path}-{split}']
            hit = [c for c in cand if os.path.exists(c + '.json')]
            if not hit:
                raise FileNotFoundError(f'no token file for split {split!r} under {path!r}')
            files.append(TokenFile(hit[0]))
    return files


def get_all_setup(model_name: str = None, model_size: str = None, model_config: Dict = None, dataset_names=None, prec: int = 5,
                  dataset_args: Dict = None, train_args: Dict = None, my_train_args: Dict = None, trainer_args: Dict = None,
                  device='cuda:0'):
    """train.py:287-368 -> (model, tokenizer, trainer).  `my_train_args` keys as the reference reads them (:299-304):
    random_crop, group_tempo, pitch_kind, insert_key, pitch_shift, channel_mixup, tokenize_scheme, tokenize_fnm,
    proportional_mixing (+ logging_strategy, save_epochs, tqdm, mode).  `dataset_args`: keys = {'train': [...], 'test': [...]}
    key name (or {name: weight}) per song for KeyInsert; seed."""
    from .data import Augment, DeviceBatcher, MixedTokenFiles
    my_train_args = dict(my_train_args or {})
    dataset_args = dict(dataset_args or {})
    names = ['random_crop', 'group_tempo', 'pitch_kind', 'insert_key', 'pitch_shift', 'channel_mixup', 'tokenize_scheme',
             'tokenize_fnm', 'proportional_mixing']
    rand_crop, grp_tp, pch_kd, ins_key, pch_shift, mix_up, tok, tok_fnm, prop_mix = (my_train_args.get(k, False) for k in names)
    tok = tok or 'vanilla'
    if tok not in ('vanilla', 'wordpiece', 'pairmerge'):
        raise ValueError(f'Tokenization Scheme mismatch: {tok!r}')
    pch_kd = pch_kd or 'midi'
    if pch_shift and not (ins_key and pch_kd == 'degree'):
        raise ValueError('A key must be inserted and the pitch kind be degree for pitch shifting')       # dataset.py:266-274
    tokenizer, model, meta = get_model_n_tokenizer(model_name, model_size, prec=prec, tokenize_scheme=tok,
                                                   tokenizer_filename=tok_fnm or None, pitch_kind=pch_kd, tempo_bin=grp_tp or None,
                                                   model_config=model_config, device=device)
    max_length = tokenizer.model_max_length
    seed = dataset_args.get('shuffle_seed', RANDOM_SEED)
    stored = MusicTokenizer(precision=prec, pitch_kind='step') if pch_shift else tokenizer     # what the token files hold

    def batcher(split: str, bsz: int):
        files = _open_split(dataset_names, split)
        if prop_mix:
            k = prop_mix if isinstance(prop_mix, int) and not isinstance(prop_mix, bool) else 2048
            tf = MixedTokenFiles(files, k if split == 'train' else max(k // 10, 1), seed=seed)    # :336-343
        elif len(files) > 1:
            tf = MixedTokenFiles(files, max(len(f) for f in files), seed=seed)
        else:
            tf = files[0]
        aug = None
        crop = bool(rand_crop) and split == 'train'                                                # dataset.py:331
        if crop or ins_key or pch_shift or mix_up:
            keys = (dataset_args.get('keys') or {}).get(split)
            aug = Augment(stored, random_crop=crop, crop_mult=rand_crop if isinstance(rand_crop, int) and rand_crop > 1 else 1,
                          insert_key=bool(ins_key), keys=keys, pitch_shift=bool(pch_shift),
                          tokenizer_degree=tokenizer if pch_shift else None, seed=seed + mdist.rank(),
                          channel_mixup=mix_up or False)
        return DeviceBatcher(tf, bsz, max_length, tokenizer.pad_token_id, device, shuffle=split == 'train', seed=seed,
                             rank=mdist.rank(), world=mdist.world_size(), augment=aug)

    pre = TrainArgs(model_name, model_size)(train_args)
    tr = batcher('train', pre['per_device_train_batch_size'])
    vl = batcher('test', pre['per_device_eval_batch_size'])
    args, my_args = get_train_and_my_train_args(model_name, model_size, train_args, my_train_args, tr)
    trainer_args = dict(trainer_args or {})
    if 'transf-xl' in model_name and not trainer_args.get('disable_train_metrics', False):
        raise NotImplementedError('train.py:364-365: additional train metrics are refused for transf-xl (GPU utilisation)')
    trainer = MyTrainer(model, tokenizer, tr, vl, train_args=args, model_name=model_name, model_size=model_size,
                        my_args=my_args, model_meta=meta, seed=seed, log_fn=trainer_args.get('log_fn'))
    return model, tokenizer, trainer


def _save_trained(trainer):
    """trainer.save_model(<out>/trained) (train.py:489,591) by rank 0 alone: the replicas are identical, and every rank writing
    the same pytorch_model.bin / trainer_state.json would race"""
    if trainer.args.get('output_dir'):
        if mdist.rank() == 0:
            trainer.save_model(os.path.join(trainer.args['output_dir'], 'trained'))
        if mdist.is_dist():
            torch.distributed.barrier()


# The reference's table of its author's training runs (musicnlp/trainer/eval.py:37-76): (model name, datasets, epochs, comment) ->
# [run directory, checkpoint directory] below <base>/models.  Data only -- the checkpoints themselves are the author's and are not
# part of any repository; the table lets the reference's `load_trained(model_key=...)` calls resolve to the same paths here.
TRAINED_KEY2PATH = {
    'full': {
        ('reformer', 'P&M', '256-256ep', 'mid-pch'): ['2022-10-03_11-58-11_reformer', 'trained'],
        ('reformer', 'All', '5-16ep', 'mid-pch_1e-4'): ['2022-10-09_01-36-18_reformer', 'checkpoint-6850'],
        ('reformer', 'All', '16-16ep', 'mid-pch_1e-4'): ['2022-10-09_01-36-18_reformer', 'trained'],
        ('reformer', 'All', 'x-128ep', '1st-prop-mix'): ['2022-10-15_22-44-10_reformer', 'trained'],
        ('transf-xl', 'All', 'x-128ep', 'prop-mix'): ['2022-10-19_04-50-21_transf-xl', 'trained'],
        ('transf-xl', 'All', '128-128ep', 'deg-pch_eval-no-mixup'): ['2022-10-26_08-41-26_transf-xl', 'trained'],
        ('transf-xl', 'All', '128-128ep', 'with-crop'): ['2022-10-27_07-56-03_transf-xl', 'trained'],
        ('transf-xl', 'All', '256-256ep', 'with-crop_train-longer'): ['2022-10-29_08-28-57_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'no-mixup'): ['2022-11-11_18-04-07_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'midi'): ['2022-11-14_13-04-30_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'midi_no-wp'): ['2022-11-18_18-22-47_transf-xl', 'checkpoint-10863'],
        ('transf-xl', 'All', '128ep', 'midi_longer-seq'): ['2022-11-21_21-22-24_transf-xl', 'checkpoint-30348'],
        ('transf-xl', 'All', '128ep', 'degree_no-wp'): ['2022-11-24_01-18-17_transf-xl', 'checkpoint-7755'],
        ('transf-xl', 'All', '128ep', 'degree_no-wp_2'): ['2022-11-24_16-29-59_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'no-wp_seg-len-512'): ['2022-11-27_13-03-40_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'large-wp'): ['2022-11-28_15-52-20_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'no-ch-mix'): ['2022-11-30_20-00-10_transf-xl', 'trained'],
        ('transf-xl', 'All', '128ep', 'small_long-seq'): ['2022-12-04_16-03-03_transf-xl', 'trained'],
    }
}


def load_trained(model_name: str = None, directory_name=None, model_key: Tuple[str, str, str, str] = None, mode: str = 'full',
                 base_path: str = None, model_dir: str = 'models', device='cuda:0'):
    """`musicnlp.trainer.eval.load_trained` (eval.py:32-95): a trained model from <base_path>/<model_dir>/<directory...>, named either
    by `directory_name` (a string or a sequence of path parts) or by a `model_key` of TRAINED_KEY2PATH (whose first element then
    also names the model).  Same argument checks: model names 'reformer' / 'transf-xl' ('transfo-xl' is accepted as well -- the
    reference validates against that spelling and then compares with the other), mode 'melody' not supported.  For
    Transformer-XL `pad_token_id = eos_token_id` is set for open-ended generation (eval.py:93-94).  `base_path` defaults to
    $MUSICNLP_BASE_PATH or the working directory (the reference's `get_base_path()` is its own checkout's parent)."""
    from .reformer import MyReformerModelWithLMHead
    from .transformer_xl import MyTransfoXLLMHeadModel
    if mode == 'melody':
        raise NotImplementedError("Current Tokenizer don't support prior melody-only representation ")
    if mode not in TRAINED_KEY2PATH:
        raise ValueError(f'Unexpected mode: expect one of {sorted(TRAINED_KEY2PATH)}, got {mode!r}')
    parts = [base_path or os.environ.get('MUSICNLP_BASE_PATH') or os.getcwd(), model_dir]
    if model_key is not None:
        model_key = tuple(model_key)
        if model_key not in TRAINED_KEY2PATH[mode]:
            raise KeyError(f'Unknown model key {model_key}: expect one of {sorted(TRAINED_KEY2PATH[mode])}')
        model_name = model_key[0]
        parts.extend(TRAINED_KEY2PATH[mode][model_key])
    elif directory_name is None:
        raise ValueError('load_trained needs a directory_name or a model_key')
    elif isinstance(directory_name, str):
        parts.append(directory_name)
    else:
        parts.extend(directory_name)
    if model_name not in ('reformer', 'transf-xl', 'transfo-xl'):
        raise ValueError(f"Unexpected Model Name: expect one of ['reformer', 'transf-xl'], got {model_name!r}")
    cls = MyReformerModelWithLMHead if model_name == 'reformer' else MyTransfoXLLMHeadModel
    model = cls.from_pretrained(os.path.join(*parts), device=device)
    if model_name != 'reformer':
        model.config.pad_token_id = model.config.eos_token_id      # for open-end generation
    return model


def train_xl(dataset_names, model_size: str = 'base', model_config: Dict = None, train_args: Dict = None,
             my_train_args: Dict = None, dataset_args: Dict = None, device='cuda:0', **train_kwargs):
    """The reference's `train_xl()` (train.py:492-593) with its hard-wired settings as defaults: max_length 1024, mem_len 512,
    cutoffs [], degree pitches with key insertion + pitch shift, crop multiple 32, proportional mixing 32768, 24 epochs, batch
    21 / eval 12, weight decay 0.1, save per epoch; seed 77.  Returns the trainer after `train()` + `save_model(<out>/trained)`."""
    debug = 'debug' in model_size
    mc = dict(max_length=1024, mem_len=512, cutoffs=[]); mc.update(model_config or {})
    ta = dict(save_strategy='epoch', num_train_epochs=24)
    if not debug:
        ta.update(weight_decay=1e-1, per_device_train_batch_size=21, per_device_eval_batch_size=12)
    ta.update(train_args or {})
    mta = dict(tqdm=True, logging_strategy='no', mode='full', random_crop=32, group_tempo=None, pitch_kind='degree',
               insert_key=True, pitch_shift=True, channel_mixup=False, tokenize_scheme='vanilla', tokenizer_filename=None,
               proportional_mixing=32768)
    mta.update(my_train_args or {})
    model, tokenizer, trainer = get_all_setup(model_name='transf-xl', model_size=model_size, model_config=mc,
                                              dataset_names=dataset_names, dataset_args=dataset_args, train_args=ta,
                                              my_train_args=mta, trainer_args=dict(disable_train_metrics=True), device=device)
    torch.manual_seed(RANDOM_SEED)
    trainer.train(**train_kwargs)
    _save_trained(trainer)
    return trainer


def train_reformer(dataset_names, model_size: str = 'base', model_config: Dict = None, train_args: Dict = None,
                   my_train_args: Dict = None, dataset_args: Dict = None, device='cuda:0', **train_kwargs):
    """The reference's `train_reformer()` (train.py:417-490): max_position_embeddings 4096 / axial (64, 64), degree pitches with
    key insertion + pitch shift, crop multiple 128, proportional mixing 1280, 64 epochs, batch 32, lr 3e-4.  (The reference runs
    it with its pair-merge tokenizer; the vanilla scheme is the default here: SURVEY 8(f) N4.)"""
    mc = dict(max_position_embeddings=4096, axial_pos_shape=(64, 64)); mc.update(model_config or {})
    ta = dict(save_strategy='epoch', num_train_epochs=64)
    if 'debug' not in model_size:
        ta.update(learning_rate=3e-4, per_device_train_batch_size=32, per_device_eval_batch_size=32)
    ta.update(train_args or {})
    mta = dict(tqdm=True, logging_strategy='no', mode='full', random_crop=128, pitch_kind='degree', insert_key=True,
               pitch_shift=True, channel_mixup=False, tokenize_scheme='vanilla', tokenizer_filename=None, proportional_mixing=1280)
    mta.update(my_train_args or {})
    model, tokenizer, trainer = get_all_setup(model_name='reformer', model_size=model_size, model_config=mc,
                                              dataset_names=dataset_names, dataset_args=dataset_args, train_args=ta,
                                              my_train_args=mta, trainer_args=dict(disable_train_metrics=True), device=device)
    trainer.train(**train_kwargs)
    _save_trained(trainer)
    return trainer
