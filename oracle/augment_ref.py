"""TEST INFRASTRUCTURE ONLY -- CPU restatement, on token STRINGS, of the reference's on-the-fly augmentations:
  * RandomCrop._crop / __call__          musicnlp/preprocess/transform.py:59-114   (the random draw `idx` is an input here)
  * KeyInsert.__call__                   musicnlp/preprocess/transform.py:138-151  (the sampled key is an input here)
  * TokenPitchShift / PitchShift         musicnlp/preprocess/transform.py:154-237  with ScaleDegreeFinder.map_single
                                         (musicnlp/preprocess/key_finder.py:245-262) and pitch_tok2midi_pitch_meta
                                         (musicnlp/vocab/music_vocab.py:712-722)
  * ChannelMixer                         musicnlp/preprocess/transform.py:331-450 with MusicConverter.str2tok_elms
                                         (musicnlp/preprocess/music_converter.py:217-274); the random draws are inputs here
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


MEL, BASS, TUP, TUP_END, EOS = '<melody>', '<bass>', '<tup>', '</tup>', '</s>'


def split_elements(toks: List[str]):
    """MusicConverter.str2tok_elms (music_converter.py:217-274): header tokens, then per bar the list of token groups -- a note
    [pitch, duration], a tuplet [<tup>, pitches..., duration, </tup>] or a lone channel marker; the trailing </s> is dropped"""
    elms, i = [], 0
    while i < len(toks):
        t = toks[i]
        if t == TUP:
            j = toks.index(TUP_END, i)
            assert j - i - 1 >= 3
            elms.append(toks[i:j + 1]); i = j + 1
        elif t.startswith('p_'):
            assert toks[i + 1].startswith('d_')
            elms.append(toks[i:i + 2]); i += 2
        else:
            elms.append([t]); i += 1
    head = [elms[0][0], elms[1][0]]
    assert head[0].startswith('TimeSig_') and head[1].startswith('Tempo_')
    elms = elms[2:]
    if elms[0][0].startswith('Key_'):
        head.append(elms[0][0]); elms = elms[1:]
    if elms[0][0] == OMIT:
        head.append(OMIT); elms = elms[1:]
    idx = [i for i, e in enumerate(elms) if e == [BAR]]
    bars = [elms[a + 1:b] for a, b in zip(idx, idx[1:] + [len(elms)])]
    if bars[-1] and bars[-1][-1] == [EOS]:
        bars[-1] = bars[-1][:-1]
    return head, bars


def channel_mix(toks: List[str], mode: str, rand, coin) -> List[str]:
    """ChannelMixer.__call__ (transform.py:351-361) + _split_bar_toks (:389-404) + _mix_up_bar_toks (:406-450).
    `rand()` -> float in [0, 1) stands for `torch.rand(1).item()` (mode 'full': one draw per interleaving decision, melody with
    probability n_melody / (n_melody + n_bass)); `coin()` -> bool for `torch.randint(2, (1,)).item() == 0` (mode 'swap': melody
    block first).  Includes the reference's quirk that a bass-only bar comes out of mode 'full' without its <bass> marker."""
    head, bars = split_elements(toks)
    out = list(head)
    for elms in bars:
        assert elms[0][0] in (MEL, BASS)
        mel, bass, cur = [], [], None
        for e in elms:
            if e[0] == MEL:
                cur = mel
            elif e[0] == BASS:
                cur = bass
            else:
                cur.append(e)
        ret = []
        if mode == 'full':
            n_m, n_b = len(mel), len(bass)
            thresh = n_m / (n_m + n_b)
            im, ib = iter(mel), iter(bass)
            em, eb = next(im, None), next(ib, None)
            prev, add_m = None, None
            while em and eb:
                add_m = rand() < thresh
                cur_marker = MEL if add_m else BASS
                if cur_marker != prev:
                    ret.append(cur_marker)
                if add_m:
                    ret += em; em = next(im, None)
                else:
                    ret += eb; eb = next(ib, None)
                prev = cur_marker
            if em:
                if not add_m:
                    ret.append(MEL)
                ret += em
                for e in im:
                    ret += e
            else:
                assert eb
                if add_m:
                    ret.append(BASS)
                ret += eb
                for e in ib:
                    ret += e
        else:
            tm = [MEL] + sum(mel, [])
            tb = [BASS] + sum(bass, [])
            ret = (tm + tb) if coin() else (tb + tm)
        out += [BAR] + ret
    out.append(EOS)
    return out
