"""TEST INFRASTRUCTURE ONLY -- CPU restatement, on token STRINGS, of the reference's on-the-fly augmentations:
  * RandomCrop._crop / __call__          musicnlp/preprocess/transform.py:59-114   (the random draw `idx` is an input here)
  * KeyInsert.__call__                   musicnlp/preprocess/transform.py:138-151  (the sampled key is an input here)
  * TokenPitchShift / PitchShift         musicnlp/preprocess/transform.py:154-237  with ScaleDegreeFinder.map_single
                                         (musicnlp/preprocess/key_finder.py:245-262) and pitch_tok2midi_pitch_meta
                                         (musicnlp/vocab/music_vocab.py:712-722)
Parity unpinned (musicnlp is not importable here; no fixtures in the reference).  Only tests/ may import this module."""
import re
from typing import List

OMIT, BAR = '[OMIT]', '<bar>'
_T0 = dict(C=0, D=1, E=2, F=3, G=4, A=5, B=6)                  # key_finder.py:199-207
_RE_STEP = re.compile(r'^p_(-?\d+)/(-?\d+)_([A-G])$')


def random_crop(toks: List[str], idx: int, min_seg_length: int = 16) -> List[str]:
    """transform.py:79-114 with the drawn bar index `idx` given (0 <= idx <= n_bar - min_seg_length); idx == 0 or a song with
    too few bars returns the song unchanged"""
    idxs_bar = [i for i, t in enumerate(toks) if t == BAR]
    if len(idxs_bar) <= min_seg_length or idx == 0:
        return list(toks)
    assert 0 <= idx <= len(idxs_bar) - min_seg_length
    global_toks = toks[:idxs_bar[0]]
    return global_toks + [OMIT] + toks[idxs_bar[idx]:]


def crop_high(toks: List[str], min_seg_length: int = 16, crop_mult: int = 1) -> int:
    """number of admissible draws - 1 (`high`, transform.py:84-91): idx = randint(0, high) * crop_mult"""
    n_bar = sum(t == BAR for t in toks)
    if n_bar <= min_seg_length:
        return 0
    high = n_bar - min_seg_length
    if crop_mult == 1:
        return high
    return high // crop_mult if high >= crop_mult else 0


def key_insert(toks: List[str], key: str) -> List[str]:
    assert toks[0].startswith('TimeSig_') and toks[1].startswith('Tempo_')
    out = list(toks)
    out.insert(2, f'Key_{key}')
    return out


def pitch_shift(toks: List[str]) -> List[str]:
    """step-pitch song with a key token at position 2 -> degree-pitch song"""
    key = toks[2]
    assert key.startswith('Key_')
    k0 = _T0[key[len('Key_')]]
    out = []
    for t in toks:
        m = _RE_STEP.match(t)
        if m is None:                       # rests, the rare-pitch token, every non-pitch token
            out.append(t)
            continue
        deg = (_T0[m.group(3)] - k0) % 7 + 1
        midi = int(m.group(1)) - 1 + (int(m.group(2)) + 1) * 12
        if midi == -12:                     # the two rare step tokens of the vocabulary (transform.py:184-191)
            midi += 12
        elif midi == 131:
            midi -= 12
        out.append(f'p_{(midi % 12) + 1}/{midi // 12 - 1}_{deg}')
    return out


def tempo_group(toks: List[str], tempo_bin: int = 5, low: int = 40, high: int = 240) -> List[str]:
    """TempoGroup.__call__ (transform.py:123-136): the tempo token at position 1 becomes the token of its bin.  Bins
    (music_vocab.py:402-417): [low, low+bin), ... from 40; the last bin also takes the high edge (one tempo longer); token
    `Tempo_{start}/{end inclusive}`; the rare-tempo tokens keep their (edge) meta and hence their string (:420-421)."""
    out = list(toks)
    tp = out[1]
    assert tp.startswith('Tempo_')
    if tp in ('Tempo_low', 'Tempo_high'):
        return out
    n = int(tp[len('Tempo_'):])
    assert (high - low) % tempo_bin == 0 and low <= n <= high
    s = low
    while s + tempo_bin <= high:
        e = s + tempo_bin                       # exclusive
        if s + tempo_bin * 2 > high:
            e += 1
        if s <= n < e:
            out[1] = f'Tempo_{s}/{e - 1}'
            return out
        s = e
    raise AssertionError(n)
