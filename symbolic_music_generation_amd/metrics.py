"""Evaluation metrics with the heavy part on the device.

Mirrors the reference's `ComputeMetrics` + `IkrMetric` (musicnlp/trainer/train.py:248-284, musicnlp/trainer/metrics.py:18-117)
and its `preprocess_logits_for_metrics=max_out_logits` hook (train.py:362):

    max_out_logits(logits)                      -> greedy ids, argmax on the GPU (mxl_argmax_rows)
    ComputeMetrics(tokenizer, mode, ...)( (preds, labels[, key_scores]) ) -> dict(ikr=..., ntp_acc=...)

The per-token work (mask, id -> pitch class, histogram, hit counting) is one kernel launch (mxl_eval_counts) that returns
14 integers per sequence; what remains on the host is the 24-key table arithmetic on a (B, 12) histogram.  The reference
moves the whole (B, T, V) logit tensor to the host for the same numbers.
"""
import re
from typing import Dict, Optional

import numpy as np
import torch

from ._lib import lib, check, MusicXLError
from .vocab import KEY_NAMES

PT_LOSS_PAD = -100

# musicnlp/vocab/elm_type.py:74-99 (`key_enum2tuple`, including its `EMin -> 'E-'` entry), :108-125, :126-129
_KEY_TUPLE = {
    'CMinor': (0, 'C'), 'C#Minor': (0, 'C#'), 'DMinor': (0, 'D'), 'EbMinor': (0, 'E-'), 'EMinor': (0, 'E-'), 'FMinor': (0, 'F'),
    'F#Minor': (0, 'F#'), 'GMinor': (0, 'G'), 'G#Minor': (0, 'G#'), 'AMinor': (0, 'A'), 'BbMinor': (0, 'B-'), 'BMinor': (0, 'B'),
    'CMajor': (1, 'C'), 'DMajor': (1, 'D'), 'DbMajor': (1, 'D-'), 'EbMajor': (1, 'E-'), 'EMajor': (1, 'E'), 'FMajor': (1, 'F'),
    'GMajor': (1, 'G'), 'GbMajor': (1, 'G-'), 'AMajor': (1, 'A'), 'AbMajor': (1, 'A-'), 'BbMajor': (1, 'B-'), 'BMajor': (1, 'B'),
}
_KEY_OFFSET = {'C': 0, 'C#': 1, 'D-': 1, 'D': 2, 'D#': 3, 'E-': 3, 'E': 4, 'F': 5, 'F#': 6, 'G-': 6, 'G': 7, 'G#': 8, 'A-': 8,
               'A': 9, 'B-': 10, 'B': 11}
_OFFKEY = {0: (1, 4, 6, 9, 11), 1: (1, 3, 6, 8, 10)}     # minor, major
_RE_PITCH = re.compile(r'^p_(-?\d+)/(-?\d+)(?:_.+)?$')


def in_key_table() -> np.ndarray:
    """(24, 12) 0/1: pitch class `pc` is in key `ordinal` (ordinal = position in `key_str2enum`, elm_type.py:44-69)"""
    tab = np.zeros((len(KEY_NAMES), 12), dtype=np.float64)
    for o, name in enumerate(KEY_NAMES):
        typ, tonic = _KEY_TUPLE[name]
        off = _KEY_OFFSET[tonic]
        for pc in range(12):
            tab[o, pc] = 0.0 if ((pc - off) % 12) in _OFFKEY[typ] else 1.0
    return tab


def pitch_class_table(vocab) -> np.ndarray:
    """int8[V]: pitch class of a pitch token id, -1 for anything else -- rests and the rare-pitch token are not pitches for the
    metric (music_tokenizer.py:94-107 with include_rest_pitch=False; pitch -> midi: music_vocab.py:582-600)"""
    tab = np.full(len(vocab), -1, dtype=np.int8)
    for i in range(len(vocab)):
        m = _RE_PITCH.match(vocab.i2t(i))
        if m:
            tab[i] = (int(m.group(1)) - 1 + (int(m.group(2)) + 1) * 12) % 12
    return tab


def pitch_class_hist_table(tokenizer) -> np.ndarray:
    """uint8 (V, 12) for a sub-word tokenizer (`id2base_ids`): how many pitches of each class one id expands to -- the reference's
    per-id pitch lists `_id2pchs_exc` (wordpiece_tokenizer.py:372-379, pair_merge_tokenizer.py:214-221) reduced to classes"""
    base = pitch_class_table(tokenizer.vocab)
    rows = tokenizer.id2base_ids()
    tab = np.zeros((len(rows), 12), dtype=np.uint8)
    for i, ids in enumerate(rows):
        for b in ids:
            if 0 <= b < len(base) and base[b] >= 0:
                tab[i, base[b]] += 1
    return tab


def max_out_logits(logits: torch.Tensor, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """train.py:362 `preprocess_logits_for_metrics`: (B, T, V) scores -> (B, T) greedy ids without leaving the device"""
    if not logits.is_cuda:
        raise MusicXLError('max_out_logits runs on the GPU only (no CPU fallback on the product path)')
    x = logits.float().contiguous()
    V = x.shape[-1]
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.int64)
    check(lib().mxl_argmax_rows(x.data_ptr(), V, out.data_ptr(), out.numel(), V, torch.cuda.current_stream().cuda_stream),
          'mxl_argmax_rows')
    return out


class ComputeMetrics:
    """`ComputeMetrics(tokenizer=, mode=, clm_pred_shifted=)` as in train.py:248-263; call with (preds, labels) for mode
    'ins-key' or (preds, labels, key_scores) for 'vanilla' (key_scores (B, 24): weight of each candidate key of the piece)."""

    def __init__(self, tokenizer, mode: str = 'vanilla', clm_pred_shifted: bool = False):
        if mode not in ('vanilla', 'ins-key'):
            raise ValueError(f'Training Mode for IKR mismatch: {mode}')
        self.tokenizer, self.vocab = tokenizer, tokenizer.vocab
        self.mode, self.clm_pred_shifted = mode, clm_pred_shifted
        # a sub-word tokenizer's ids expand to several base tokens: per-id pitch-class counts instead of one class per id
        self.subword = hasattr(tokenizer, 'id2base_ids')
        self._pc_host = pitch_class_hist_table(tokenizer) if self.subword else pitch_class_table(self.vocab)
        self._pc_dev: Dict[torch.device, torch.Tensor] = {}
        self._in_key = in_key_table()
        self._key_id = {self.vocab.t2i(f'Key_{k}'): o for o, k in enumerate(KEY_NAMES) if f'Key_{k}' in self.vocab}
        if self.subword:        # the key token of 'ins-key' labels, as an id of the sub-word vocabulary (ids of one base token)
            base_key = self._key_id
            self._key_id = {i: base_key[b[0]] for i, b in enumerate(tokenizer.id2base_ids()) if len(b) == 1 and b[0] in base_key}

    def counts(self, preds: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """(B, 14) int32 on the device: pitch-class histogram [12], #correct, #counted"""
        if not preds.is_cuda:
            raise MusicXLError('ComputeMetrics runs on the GPU only (no CPU fallback on the product path)')
        preds, labels = preds.to(torch.int64).contiguous(), labels.to(torch.int64).contiguous()
        B, T = labels.shape
        want = T - 1 if self.clm_pred_shifted else T
        if preds.shape != (B, want):
            raise ValueError(f'Input and label shapes do not match, {tuple(preds.shape)} vs {tuple(labels.shape)}')
        if preds.device not in self._pc_dev:
            self._pc_dev[preds.device] = torch.from_numpy(self._pc_host).to(preds.device)
        out = torch.empty(B, 14, device=preds.device, dtype=torch.int32)
        fn, what = (lib().mxl_eval_counts_multi, 'mxl_eval_counts_multi') if self.subword else (lib().mxl_eval_counts, 'mxl_eval_counts')
        check(fn(preds.data_ptr(), preds.stride(0), labels.data_ptr(), labels.stride(0), self._pc_dev[preds.device].data_ptr(),
                 len(self._pc_host), out.data_ptr(), B, T, int(self.clm_pred_shifted), torch.cuda.current_stream().cuda_stream), what)
        return out

    def ikr_from_counts(self, counts: np.ndarray, labels=None, key_scores=None) -> float:
        hist = counts[:, :12].astype(np.float64)
        n = hist.sum(axis=1)
        ratio = np.where(n[:, None] > 0, hist @ self._in_key.T / np.maximum(n, 1)[:, None], 0.0)     # (B, 24); no pitch -> 0
        if self.mode == 'vanilla':
            w = np.asarray(key_scores, dtype=np.float64)
            w = np.where(w > 0, w, 0.0)
            per_row = (ratio * w).sum(axis=1) / w.sum(axis=1)
        else:
            lab = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
            pos = 2                      # the key token is the 3rd label token (metrics.py:63)
            per_row = np.empty(len(ratio))
            for b in range(len(ratio)):
                kid = int(lab[b, pos])
                if kid not in self._key_id:
                    raise ValueError(f'Expect key token at 3rd position of label, got {kid}')
                per_row[b] = ratio[b, self._key_id[kid]]
        return float(per_row.mean())

    def __call__(self, eval_pred) -> Dict[str, float]:
        if self.mode == 'vanilla':
            preds, labels, key_scores = eval_pred
        else:
            (preds, labels), key_scores = eval_pred, None
        c = self.counts(preds, labels).cpu().numpy()
        if isinstance(key_scores, torch.Tensor):
            key_scores = key_scores.cpu().numpy()
        return dict(ikr=self.ikr_from_counts(c, labels, key_scores),
                    ntp_acc=float(c[:, 12].sum()) / max(float(c[:, 13].sum()), 1.0))
