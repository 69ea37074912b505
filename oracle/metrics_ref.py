"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation metrics.

Follows, line by line:
  * `IkrMetric.__call__` / `get_in_key_ratio`        musicnlp/trainer/metrics.py:45-69, 103-117
  * `ComputeMetrics.__call__` (ntp_acc)              musicnlp/trainer/train.py:265-284
  * key tables                                        musicnlp/vocab/elm_type.py:31-131  (including its `EMin -> 'E-'` entry,
                                                      kept as is: the metric is defined by the table, not by music theory)
  * `MusicTokenizer.ids2pitches`                      musicnlp/vocab/music_tokenizer.py:94-107
  * pitch token -> midi                               musicnlp/vocab/music_vocab.py:582-607

Parity unpinned: `musicnlp` cannot be imported here (music21 / stefutil absent) and the reference holds no metric goldens;
the anchors are hand-computed cases in tests/test_metrics_cpu.py.
Only tests/ may import this module."""
import re
from typing import List, Optional, Sequence

import numpy as np

PT_LOSS_PAD = -100

# elm_type.py:44-69: ordinal = position in `key_str2enum`
KEY_STRS = ['CMajor', 'FMajor', 'BbMajor', 'EbMajor', 'AbMajor', 'DbMajor', 'GbMajor', 'BMajor', 'EMajor', 'AMajor', 'DMajor',
            'GMajor', 'AMinor', 'DMinor', 'GMinor', 'CMinor', 'FMinor', 'BbMinor', 'EbMinor', 'G#Minor', 'C#Minor', 'F#Minor',
            'BMinor', 'EMinor']
# elm_type.py:74-99 `key_enum2tuple`: (type: 1 major / 0 minor, tonic name)
_KEY_TUPLE = {
    'CMinor': (0, 'C'), 'C#Minor': (0, 'C#'), 'DMinor': (0, 'D'), 'EbMinor': (0, 'E-'), 'EMinor': (0, 'E-'), 'FMinor': (0, 'F'),
    'F#Minor': (0, 'F#'), 'GMinor': (0, 'G'), 'G#Minor': (0, 'G#'), 'AMinor': (0, 'A'), 'BbMinor': (0, 'B-'), 'BMinor': (0, 'B'),
    'CMajor': (1, 'C'), 'DMajor': (1, 'D'), 'DbMajor': (1, 'D-'), 'EbMajor': (1, 'E-'), 'EMajor': (1, 'E'), 'FMajor': (1, 'F'),
    'GMajor': (1, 'G'), 'GbMajor': (1, 'G-'), 'AMajor': (1, 'A'), 'AbMajor': (1, 'A-'), 'BbMajor': (1, 'B-'), 'BMajor': (1, 'B'),
}
# elm_type.py:108-125
_KEY_OFFSET = {'C': 0, 'C#': 1, 'D-': 1, 'D': 2, 'D#': 3, 'E-': 3, 'E': 4, 'F': 5, 'F#': 6, 'G-': 6, 'G': 7, 'G#': 8, 'A-': 8,
               'A': 9, 'B-': 10, 'B': 11}
_OFFKEY = [[1, 4, 6, 9, 11], [1, 3, 6, 8, 10]]          # elm_type.py:126-129: [minor, major]

_RE_PITCH = re.compile(r'^p_(-?\d+)/(-?\d+)(?:_.+)?$')


def key_type_offset(ordinal: int):
    typ, name = _KEY_TUPLE[KEY_STRS[ordinal]]
    return typ, _KEY_OFFSET[name]


def ids2pitches(tokens: Sequence[str]) -> List[int]:
    """midi numbers of the pitch tokens, rests (`p_r`) and the rare-pitch token excluded (include_rest_pitch=False)"""
    out = []
    for t in tokens:
        m = _RE_PITCH.match(t)
        if m:
            out.append(int(m.group(1)) - 1 + (int(m.group(2)) + 1) * 12)
    return out


def in_key_ratio(pitches: Sequence[int], ordinal: int) -> float:
    if len(pitches) == 0:                      # metrics.py:107-108
        return 0.0
    typ, off = key_type_offset(ordinal)
    rel = [((p % 12) - off) % 12 for p in pitches]
    return sum(x not in _OFFKEY[typ] for x in rel) / len(pitches)


def ikr(preds: np.ndarray, labels: np.ndarray, id2tok, key_scores: Optional[np.ndarray] = None, mode: str = 'vanilla',
        clm_pred_shifted: bool = False) -> float:
    if clm_pred_shifted:
        labels = labels[:, 1:]
    assert preds.shape == labels.shape
    vals = []
    for b in range(preds.shape[0]):
        pred = preds[b][labels[b] != PT_LOSS_PAD]
        pitches = ids2pitches([id2tok(int(i)) for i in pred])
        if mode == 'vanilla':
            ords = [o for o in range(len(key_scores[b])) if key_scores[b][o] > 0]
            w = [float(key_scores[b][o]) for o in ords]
            vals.append(float(np.average([in_key_ratio(pitches, o) for o in ords], weights=w)))
        else:                                  # 'ins-key': the key token sits at label position 2 (1 when shifted)
            tok = id2tok(int(labels[b][1 if clm_pred_shifted else 2]))
            if not tok.startswith('Key_'):
                raise ValueError(f'Expect key token at 3rd position of label, got {tok}')
            vals.append(in_key_ratio(pitches, KEY_STRS.index(tok[len('Key_'):])))
    return float(np.mean(vals))


def ntp_acc(preds: np.ndarray, labels: np.ndarray, clm_pred_shifted: bool = False) -> float:
    if not clm_pred_shifted:
        preds = preds[:, :-1]
    labels = labels[:, 1:]
    labels, preds = labels.flatten(), preds.flatten()
    msk = labels != PT_LOSS_PAD
    return float((preds[msk] == labels[msk]).mean())
