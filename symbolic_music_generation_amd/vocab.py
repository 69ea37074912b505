"""Drop-in `MusicVocabulary` / `MusicTokenizer` (reference: musicnlp/vocab/music_vocab.py:112-951, music_tokenizer.py:15-110)
restated in pure Python: no music21, no HF `PreTrainedTokenizer`.  String <-> id on the CPU is not on the accelerated path
(<< 1 % of a step); what matters is that the id layout is the reference's, because it fixes vocab_size (422 / 560 / 1190
for pitch kind midi / step / degree) and therefore the adaptive-softmax cutoffs and every embedding row.

Id layout (music_vocab.py:349-375): special(8) -> time_sig -> tempo -> key -> pitch -> duration, consecutive.
"""
import re
from collections import OrderedDict
from fractions import Fraction
from typing import Dict, Iterable, List, Optional, Sequence, Union

import math

COMMON_TIME_SIGS = sorted([(4, 4), (2, 4), (2, 2), (3, 4), (6, 8), (5, 4), (12, 8)], key=lambda t: tuple(reversed(t)))  # :29-32
TEMPO_LOW_EDGE, TEMPO_HIGH_EDGE = 40, 240                                                                              # :33
COMMON_TEMPOS = list(range(TEMPO_LOW_EDGE, TEMPO_HIGH_EDGE + 1))

KEY_NAMES = [  # musicnlp/vocab/elm_type.py:46-71
    'CMajor', 'FMajor', 'BbMajor', 'EbMajor', 'AbMajor', 'DbMajor', 'GbMajor', 'BMajor', 'EMajor', 'AMajor', 'DMajor',
    'GMajor', 'AMinor', 'DMinor', 'GMinor', 'CMinor', 'FMinor', 'BbMinor', 'EbMinor', 'G#Minor', 'C#Minor', 'F#Minor',
    'BMinor', 'EMinor',
]

# step names per atonal pitch index (normal + rare), music_vocab.py:190-205
_ATONAL = {
    1: (['C'], ['B#']), 2: (['C#', 'D-'], []), 3: (['D'], ['C##']), 4: (['D#', 'E-'], []), 5: (['E'], ['F-']),
    6: (['F'], ['E#']), 7: (['F#', 'G-'], []), 8: (['G'], ['F##']), 9: (['G#', 'A-'], []), 10: (['A'], ['B--', 'G##']),
    11: (['A#', 'B-'], []), 12: (['B'], ['C-']),
}


class MusicVocabulary:
    pad = '[PAD]'
    omitted_segment = '[OMIT]'
    start_of_bar = '<bar>'
    start_of_melody = '<melody>'
    start_of_bass = '<bass>'
    end_of_song = '</s>'
    start_of_tuplet = '<tup>'
    end_of_tuplet = '</tup>'
    sep = '_'
    rare_time_sig = 'TimeSig_rare'
    rare_low_tempo = 'Tempo_low'
    rare_high_tempo = 'Tempo_high'
    rare_pitch = 'p_rare'
    rare_duration = 'd_rare'
    rest = 'p_r'

    def __init__(self, precision: int = 5, color: bool = False, is_wordpiece: bool = False, pitch_kind: str = 'midi',
                 with_rare_step: bool = True, tempo_bin: Union[bool, int, None] = None):
        if pitch_kind not in ('midi', 'step', 'degree'):
            raise ValueError(f'Unique Pitch Kind mismatch: {pitch_kind}')
        # `is_wordpiece` only changes how the reference colours tokens on a terminal (music_vocab.py:754-756): ids are the same
        self.precision, self.pitch_kind, self.with_rare_step = precision, pitch_kind, with_rare_step
        self.is_wordpiece = is_wordpiece
        self.tempo_bin = (5 if tempo_bin is True else tempo_bin) if tempo_bin else None
        special = [self.omitted_segment, self.pad, self.start_of_bar, self.end_of_song, self.start_of_melody,
                   self.start_of_bass, self.start_of_tuplet, self.end_of_tuplet]                                  # :358-362
        tss = [f'TimeSig_{num}/{den}' for (den, num) in sorted((den, num) for (num, den) in COMMON_TIME_SIGS)]
        keys = [f'Key_{k}' for k in sorted(KEY_NAMES)]
        self.toks: "OrderedDict[str, List[str]]" = OrderedDict(
            special=special,
            time_sig=[self.rare_time_sig, *tss],
            tempo=[self.rare_low_tempo, *self._tempos(), self.rare_high_tempo],
            key=keys,
            pitch=self._pitches(),
            duration=[self.rare_duration, *self.get_durations()],
        )
        for toks in self.toks.values():
            assert len(set(toks)) == len(toks)
        self.tok2id: Dict[str, int] = {}
        for toks in self.toks.values():
            for t in toks:
                self.tok2id[t] = len(self.tok2id)
        self.id2tok = {v: k for k, v in self.tok2id.items()}
        self._re_pitch = re.compile(r'^p_(-?\d+)/(-?\d+)(?:_([A-G1-7]))?$')
        self._re_tempo = re.compile(r'^Tempo_(-?\d+)$')

    # ---- construction helpers
    def _tempos(self) -> List[str]:
        if not self.tempo_bin:
            return [f'Tempo_{t}' for t in COMMON_TEMPOS]
        assert (TEMPO_HIGH_EDGE - TEMPO_LOW_EDGE) % self.tempo_bin == 0
        out, s = [], TEMPO_LOW_EDGE
        while s + self.tempo_bin <= TEMPO_HIGH_EDGE:                                                               # :405-420
            e = s + self.tempo_bin
            if s + self.tempo_bin * 2 > TEMPO_HIGH_EDGE:
                e += 1
            out.append(f'Tempo_{s}/{e - 1}')
            s = e
        return out

    def _pitches(self) -> List[str]:
        ret = [self.rest, self.rare_pitch]                                                                          # :443
        if self.pitch_kind == 'midi':
            ret += [f'p_{(i % 12) + 1}/{i // 12 - 1}' for i in range(128)]
        elif self.pitch_kind == 'degree':
            ret += [f'p_{(i % 12) + 1}/{i // 12 - 1}_{dgr}' for i in range(128) for dgr in range(1, 8)]
        else:
            for i in range(128):
                idx = (i % 12) + 1
                normal, rare = _ATONAL[idx]
                for name in (normal + rare if self.with_rare_step else normal):
                    otv = i // 12 - 1
                    if idx == 1 and name == 'B#':
                        otv -= 1
                    elif idx == 12 and name == 'C-':
                        otv += 1
                    ret.append(f'p_{idx}/{otv}_{name[0]}')
        assert len(ret) == len(set(ret))
        return ret

    def get_durations(self) -> List[str]:
        bound = max(n / dd for n, dd in COMMON_TIME_SIGS) * 4                                                       # :506
        slot = Fraction(4, 2 ** self.precision)
        out = []
        for i in range(math.ceil(bound / slot)):
            f = (i + 1) * slot
            out.append(f'd_{f.numerator}' if f.denominator == 1 else f'd_{f.numerator}/{f.denominator}')
        return out

    # ---- lookup
    def __len__(self):
        return len(self.tok2id)

    def __contains__(self, tok):
        return tok in self.tok2id

    def type(self, tok: str) -> str:
        if tok.startswith('p_'):
            return 'pitch'
        if tok.startswith('d_'):
            return 'duration'
        if tok.startswith('TimeSig_'):
            return 'time_sig'
        if tok.startswith('Tempo_'):
            return 'tempo'
        if tok.startswith('Key_'):
            return 'key'
        return 'special'

    def sanitize_rare_token(self, tok: str, for_midi: bool = False) -> str:
        """music_vocab.py:883-915: out-of-vocabulary tokens map to their type's rare token."""
        if tok in self.tok2id:
            return tok
        typ = self.type(tok)
        if typ == 'pitch':
            if for_midi:
                m = self._re_pitch.match(tok)
                mid = int(m.group(1)) - 1 + (int(m.group(2)) + 1) * 12
                while mid < 0:
                    mid += 12
                while mid > 127:
                    mid -= 12
                return f'p_{(mid % 12) + 1}/{mid // 12 - 1}'
            return self.rare_pitch
        if typ == 'duration':
            return self.rare_duration
        if typ == 'time_sig':
            return self.rare_time_sig
        if typ == 'tempo':
            m = self._re_tempo.match(tok)
            if not m:
                raise KeyError(tok)
            return self.rare_low_tempo if int(m.group(1)) < 40 else self.rare_high_tempo
        raise KeyError(f'unknown token {tok!r}')

    def t2i(self, tok: str) -> int:                                                                                 # :924-926
        return self.tok2id[self.sanitize_rare_token(tok)]

    def i2t(self, id_: int) -> str:
        return self.id2tok[int(id_)]


class MusicTokenizer:
    """Whitespace split -> `t2i`; `model_input_names = ['input_ids']` so no attention mask is ever produced
    (music_tokenizer.py:23, 81-82).  Implements the slice of the HF tokenizer API the reference's callers use."""
    model_input_names = ['input_ids']

    def __init__(self, precision: int = 5, is_wordpiece: bool = False, pitch_kind: str = 'midi',
                 tempo_bin: Union[bool, int, None] = None, vocab: Optional[MusicVocabulary] = None, model_max_length: int = 4096):
        self.precision, self.is_wordpiece, self.pitch_kind = precision, is_wordpiece, pitch_kind
        self.vocab = vocab or MusicVocabulary(precision=precision, is_wordpiece=is_wordpiece, pitch_kind=pitch_kind,
                                              tempo_bin=tempo_bin)
        self.model_max_length = model_max_length
        self.pad_token, self.eos_token = self.vocab.pad, self.vocab.end_of_song                                    # :52-53
        self.sob_token = self.vocab.start_of_bar
        self.pad_token_id = self.vocab.t2i(self.pad_token)
        self.eos_token_id = self.vocab.t2i(self.eos_token)
        self.sob_token_id = self.vocab.t2i(self.sob_token)
        self.padding_side = 'right'

    @property
    def vocab_size(self) -> int:
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab)

    def tokenize(self, text: str) -> List[str]:
        return text.split()

    def convert_tokens_to_ids(self, toks: Union[str, Sequence[str]]):
        return self.vocab.t2i(toks) if isinstance(toks, str) else [self.vocab.t2i(t) for t in toks]

    def convert_ids_to_tokens(self, ids: Union[int, Iterable[int]]):
        return self.vocab.i2t(ids) if isinstance(ids, int) else [self.vocab.i2t(i) for i in ids]

    def encode(self, text: str) -> List[int]:
        return self.convert_tokens_to_ids(self.tokenize(text))

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        ids = ids.tolist() if hasattr(ids, 'tolist') else list(ids)
        toks = [self.vocab.i2t(i) for i in ids]
        if skip_special_tokens:
            toks = [t for t in toks if t != self.pad_token]
        return ' '.join(toks)

    def __call__(self, text, padding=False, truncation=False, max_length: Optional[int] = None, return_tensors=None):
        texts = [text] if isinstance(text, str) else list(text)
        enc = [self.encode(t) for t in texts]
        max_length = max_length or self.model_max_length
        if truncation:
            enc = [e[:max_length] for e in enc]
        if padding == 'max_length':
            enc = [e + [self.pad_token_id] * (max_length - len(e)) for e in enc]
        elif padding in (True, 'longest'):
            m = max(len(e) for e in enc)
            enc = [e + [self.pad_token_id] * (m - len(e)) for e in enc]
        if return_tensors == 'pt':
            import torch
            return {'input_ids': torch.tensor(enc, dtype=torch.long)}
        return {'input_ids': enc[0] if isinstance(text, str) else enc}
