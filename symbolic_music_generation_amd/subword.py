"""Sub-word tokenizers of the reference, restated over this package's pure-Python `MusicVocabulary`:

* `PairMergeTokenizerTrainer` / `PairMergeTokenizer` -- musicnlp/trainer/pair_merge_tokenizer.py:29-289: every note `pitch dur`
  and every tuplet `<tup> p.. dur </tup>` is one *music element*; the most frequent elements of a corpus become single merged
  tokens appended to the vocabulary (ids = original size + rank), a song then has exactly one tokenization.
* `Score2Chars` / `WordPieceMusicTokenizerTrainer` / `WordPieceMusicTokenizer` -- musicnlp/trainer/wordpiece_tokenizer.py:28-452:
  every vocabulary token maps to one unicode character, "words" are the stretches between bar / channel / tuplet markers
  (`punctuate`) with the global tokens kept apart (`independent_global_token`), and a WordPiece model (HuggingFace
  `tokenizers`, the library the reference uses) learns merges inside words.

Both keep the `MusicTokenizer` call surface (`__call__` with padding / truncation, `vocab_size`, pad / eos ids, `decode`,
`ids2pitches`), so `get_model_n_tokenizer(tokenize_scheme=...)` can hand them to the model configs: a vocabulary of 1000 ids or
more switches `MyTransfoXLConfig` to an adaptive softmax with cutoffs (musicnlp/models/transformer_xl.py:53-66), which the HIP
head kernels cover.  Host-side string work: none of this is on the device path.
"""
import json
from collections import Counter
from typing import Dict, Iterable, List, Optional, Sequence, Union

from .vocab import MusicTokenizer, MusicVocabulary

__all__ = ['split_song', 'PairMergeTokenizerTrainer', 'PairMergeTokenizer', 'Score2Chars', 'WordPieceMusicTokenizerTrainer',
           'WordPieceMusicTokenizer']


class SongSplit:
    __slots__ = ('time_sig', 'tempo', 'key', 'omit', 'elms_by_bar', 'end_of_song')


def split_song(vocab: MusicVocabulary, text: Union[str, Sequence[str]]) -> SongSplit:
    """`MusicConverter.str2tok_elms` (musicnlp/preprocess/music_converter.py:217-274): header tokens and, per bar, the token
    groups -- note [pitch, duration], tuplet [<tup>, pitches.., duration, </tup>], lone channel marker"""
    toks = text.split() if isinstance(text, str) else list(text)
    elms, i = [], 0
    while i < len(toks):
        t = toks[i]
        if t == vocab.start_of_tuplet:
            j = toks.index(vocab.end_of_tuplet, i)
            if j - i - 1 < 3:
                raise ValueError('a tuplet holds at least two pitches and a duration')
            elms.append(toks[i:j + 1]); i = j + 1
        elif vocab.type(t) == 'pitch':
            if i + 1 >= len(toks) or vocab.type(toks[i + 1]) != 'duration':
                raise ValueError(f'pitch {t!r} without a duration')
            elms.append(toks[i:i + 2]); i += 2
        else:
            elms.append([t]); i += 1
    out = SongSplit()
    out.time_sig, out.tempo, out.key, out.omit = elms[0][0], elms[1][0], None, None
    if vocab.type(out.time_sig) != 'time_sig' or vocab.type(out.tempo) != 'tempo':
        raise ValueError('a song opens with its time signature and tempo')
    elms = elms[2:]
    if elms and vocab.type(elms[0][0]) == 'key':
        out.key, elms = elms[0][0], elms[1:]
    if elms and elms[0][0] == vocab.omitted_segment:
        out.omit, elms = elms[0][0], elms[1:]
    idx = [k for k, e in enumerate(elms) if e == [vocab.start_of_bar]]
    bars = [elms[a + 1:b] for a, b in zip(idx, idx[1:] + [len(elms)])]
    out.end_of_song = None
    if bars and bars[-1] and bars[-1][-1] == [vocab.end_of_song]:
        bars[-1] = bars[-1][:-1]
        out.end_of_song = vocab.end_of_song
    out.elms_by_bar = bars
    return out


# --------------------------------------------------------------------------------------------------------------- pair merge
class PairMergeTokenizerTrainer:
    """pair_merge_tokenizer.py:29-196.  `songs`: an iterable of token strings (the reference reads them from its datasets and,
    for degree pitches, expands every song over its candidate keys first -- pass the expanded corpus)."""

    def __init__(self, pitch_kind: str = 'step', precision: int = 5, **kwargs):
        self.pitch_kind = pitch_kind
        self.vocab = MusicVocabulary(pitch_kind=pitch_kind, precision=precision, **kwargs)

    def song2elements(self, song) -> List[str]:
        v = self.vocab
        return [' '.join(me) for elms in split_song(v, song).elms_by_bar for me in elms
                if me != [v.start_of_melody] and me != [v.start_of_bass]]

    def __call__(self, songs: Iterable, vocab_size: Optional[int] = None, coverage_ratio: Optional[float] = None,
                 save: Optional[str] = None) -> Dict:
        if bool(vocab_size) == bool(coverage_ratio):
            raise ValueError('Specify one of vocab_size, coverage_ratio')
        c, n = Counter(), 0
        for s in songs:
            c.update(self.song2elements(s)); n += 1
        n_uniq = len(c)
        mc_all = c.most_common()
        total = sum(v for _, v in mc_all)
        cum, ratio = 0, []
        for _, v in mc_all:
            cum += v; ratio.append(cum / total)
        if vocab_size:
            vsz_add = vocab_size - len(self.vocab)
            if vsz_add > n_uniq:                                      # :101-105
                vsz_add, coverage_ratio = n_uniq, 1.0
            else:
                coverage_ratio = ratio[vsz_add] if vsz_add < len(ratio) else 1.0
        else:
            import bisect
            vsz_add = bisect.bisect_right(ratio, coverage_ratio)      # np.searchsorted(ratio, r, side='right')
            vocab_size = len(self.vocab) + vsz_add
        mc = mc_all[:vsz_add]
        n_vocab = len(self.vocab)
        meta = dict(added_tok2id={tok: i + n_vocab for i, (tok, _) in enumerate(mc)},          # descending frequency (:118-120)
                    n_unique=n_uniq, n_added=vsz_add, occurence_count=dict(mc), original_vocab_size=n_vocab,
                    music_vocab=dict(precision=self.vocab.precision, pitch_kind=self.vocab.pitch_kind),
                    coverage_ratio=coverage_ratio, n_songs=n)
        if save:
            with open(save if save.endswith('.json') else save + '.json', 'w') as f:
                json.dump(meta, f, indent=4)
        return meta


class PairMergeTokenizer(MusicTokenizer):
    """pair_merge_tokenizer.py:199-289"""

    def __init__(self, added_tok2id: Dict[str, int], precision: int = 5, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self.original_vocab_size = len(self.vocab)
        self.added_tok2id = dict(added_tok2id)
        self.added_id2tok = {v: k for k, v in self.added_tok2id.items()}
        self.added_vocab_size = len(self.added_tok2id)
        if sorted(self.added_id2tok) != list(range(self.original_vocab_size, self.original_vocab_size + self.added_vocab_size)):
            raise ValueError('merged tokens must take the ids right after the base vocabulary')

    @classmethod
    def from_file(cls, path: str, **kwargs):
        with open(path if path.endswith('.json') else path + '.json') as f:
            meta = json.load(f)
        mv = dict(meta['music_vocab'])
        if 'precison' in mv:                      # the reference's files spell it this way (pair_merge_tokenizer.py:124)
            mv['precision'] = mv.pop('precison')
        ret = cls(added_tok2id=meta['added_tok2id'], **{**mv, **kwargs})
        if meta['original_vocab_size'] != len(ret.vocab):
            raise ValueError('the tokenizer file was trained on a different base vocabulary')
        return ret

    @property
    def vocab_size(self) -> int:
        return self.original_vocab_size + self.added_vocab_size

    def __len__(self):
        return self.vocab_size

    def tokenize(self, text) -> List[str]:
        out = split_song(self.vocab, text)
        ret = [out.time_sig, out.tempo]
        if out.key:
            ret.append(out.key)
        if out.omit:
            ret.append(out.omit)
        for elms in out.elms_by_bar:
            ret.append(self.sob_token)
            for me in elms:
                merged = ' '.join(me)
                if merged in self.added_tok2id:
                    ret.append(merged)
                else:
                    ret += me
        if out.end_of_song:
            ret.append(out.end_of_song)
        return ret

    def _convert_token_to_id(self, token: str) -> int:
        return self.added_tok2id[token] if token in self.added_tok2id else self.vocab.t2i(token)

    def _convert_id_to_token(self, index: int) -> str:
        return self.vocab.i2t(index) if index < self.original_vocab_size else self.added_id2tok[int(index)]

    def convert_tokens_to_ids(self, toks):
        return self._convert_token_to_id(toks) if isinstance(toks, str) else [self._convert_token_to_id(t) for t in toks]

    def convert_ids_to_tokens(self, ids):
        return self._convert_id_to_token(ids) if isinstance(ids, int) else [self._convert_id_to_token(int(i)) for i in ids]

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        ids = ids.tolist() if hasattr(ids, 'tolist') else list(ids)
        toks = [self._convert_id_to_token(int(i)) for i in ids]
        if skip_special_tokens:
            toks = [t for t in toks if t != self.pad_token]
        return ' '.join(toks)

    def id2base_ids(self) -> List[List[int]]:
        """per id, the base-vocabulary ids it stands for (what the device-side metric tables are built from)"""
        return [[self.vocab.t2i(t) for t in self._convert_id_to_token(i).split()] for i in range(self.vocab_size)]


# --------------------------------------------------------------------------------------------------------------- word piece
def _uni_chars() -> List[str]:
    """wordpiece_tokenizer.py:28-53: a fixed, sorted list of printing-friendly unicode characters"""
    ranges = [(0x0021, 0x02FF), (0x0080, 0x00FF), (0x0100, 0x017F), (0x0180, 0x024F), (0x0250, 0x02AF), (0x1D00, 0x1D7F),
              (0x1D80, 0x1DBF), (0x1E00, 0x1EFF), (0x2100, 0x214F)]
    omit = set(range(0x7f, 0xa1)) | {0xad}
    return sorted({chr(i) for a, b in ranges for i in range(a, b) if i not in omit})


class Score2Chars:
    """wordpiece_tokenizer.py:56-250: music tokens <-> characters, words separated by blanks"""
    uni_chars_cache = _uni_chars()

    def __init__(self, vocab: MusicVocabulary, chars: Optional[List[str]] = None, continuing_prefix: str = '##',
                 independent_global_token: bool = False, punctuate: bool = False, omit_eos: bool = False):
        self.vocab = vocab
        if chars is None:
            if len(vocab) > len(self.uni_chars_cache):
                raise ValueError('vocabulary larger than the character table')
            chars = self.uni_chars_cache[:len(vocab)]
        assert len(chars) == len(vocab) and all(c != ' ' for c in chars)
        self.dec_chars = list(chars)
        self.enc_chars = {c: i for i, c in enumerate(chars)}
        self.continuing_prefix = continuing_prefix
        self.independent_global_token, self.punctuate, self.omit_eos = independent_global_token, punctuate, omit_eos
        self.need_split = independent_global_token or punctuate
        v = vocab
        self.spec_toks = {v.start_of_bar, v.start_of_tuplet, v.end_of_tuplet, v.end_of_song, v.start_of_melody, v.start_of_bass}

    def split(self, score) -> List[List[str]]:
        """-> words (lists of tokens), wordpiece_tokenizer.py:137-185"""
        toks = score.split() if isinstance(score, str) else list(score)
        if not self.need_split:
            return [toks]
        v = self.vocab
        ts, tp, rest = toks[0], toks[1], toks[2:]
        key = omit = None
        if rest and v.type(rest[0]) == 'key':
            key, rest = rest[0], rest[1:]
        if rest and rest[0] == v.omitted_segment:
            omit, rest = rest[0], rest[1:]
        assert rest[0] == v.start_of_bar and (self.omit_eos or rest[-1] == v.end_of_song)
        if self.independent_global_token:
            words = [[ts], [tp]] + ([[key]] if key else []) + ([[omit]] if omit else [])
            return words + (self._split_bar_notes(rest) if self.punctuate else [rest])
        head = [ts, tp] + ([key] if key else []) + ([omit] if omit else [])
        return [head] + self._split_bar_notes(rest)

    def _split_bar_notes(self, toks: List[str]) -> List[List[str]]:
        words, cur = [], []
        for t in toks:
            if t in self.spec_toks:
                if cur:
                    words.append(cur)
                words.append([t]); cur = []
            else:
                cur.append(t)
        if cur:
            words.append(cur)
        return words

    def encode_single(self, toks) -> str:
        toks = toks.split() if isinstance(toks, str) else toks
        return ''.join(self.dec_chars[self.vocab.t2i(t)] for t in toks)

    def __call__(self, score) -> str:
        return ' '.join(self.encode_single(w) for w in self.split(score))

    def decode_single(self, s: str) -> str:
        s = s[len(self.continuing_prefix):] if s.startswith(self.continuing_prefix) else s
        return ' '.join(self.vocab.i2t(self.enc_chars[c]) for c in s)

    def decode(self, s: str) -> str:
        return ' '.join(self.decode_single(w) for w in s.split())


class WordPieceMusicTokenizerTrainer:
    """wordpiece_tokenizer.py:253-345: a `tokenizers` WordPiece model trained on the character form of a corpus"""

    def __init__(self, vocab: Optional[MusicVocabulary] = None, pitch_kind: str = 'midi', precision: int = 5,
                 independent_global_token: bool = True, punctuate: bool = True, continuing_prefix: str = '##', **kwargs):
        self.vocab = vocab or MusicVocabulary(pitch_kind=pitch_kind, precision=precision, is_wordpiece=True, **kwargs)
        self.s2c = Score2Chars(self.vocab, continuing_prefix=continuing_prefix,
                               independent_global_token=independent_global_token, punctuate=punctuate)

    def __call__(self, songs: Iterable, vocab_size: int = 8192, save: Optional[str] = None) -> 'WordPieceMusicTokenizer':
        from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
        unk = self.s2c.encode_single([self.vocab.pad])
        tok = Tokenizer(models.WordPiece(unk_token=unk, max_input_chars_per_word=int(1e10)))
        tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
        tok.decoder = decoders.WordPiece(prefix=self.s2c.continuing_prefix)
        trainer = trainers.WordPieceTrainer(vocab_size=vocab_size, initial_alphabet=self.s2c.dec_chars, show_progress=False,
                                            continuing_subword_prefix=self.s2c.continuing_prefix, special_tokens=[unk])
        tok.train_from_iterator((self.s2c(s) for s in songs), trainer=trainer)
        s2c_args = dict(independent_global_token=self.s2c.independent_global_token, punctuate=self.s2c.punctuate)
        if save:
            tok.save(save + '.json')
            with open(save + '_meta.json', 'w') as f:
                json.dump(dict(music_vocab=dict(precision=self.vocab.precision, pitch_kind=self.vocab.pitch_kind),
                               score2chars=s2c_args, tok2id=self.vocab.tok2id), f, indent=4)
        return WordPieceMusicTokenizer(tok, precision=self.vocab.precision, pitch_kind=self.vocab.pitch_kind, s2c_args=s2c_args)


class WordPieceMusicTokenizer(MusicTokenizer):
    """wordpiece_tokenizer.py:349-452 over a trained `tokenizers.Tokenizer`"""

    def __init__(self, tokenizer, precision: int = 5, s2c_args: Optional[Dict] = None, omit_eos: bool = False, **kwargs):
        super().__init__(precision=precision, is_wordpiece=True, **kwargs)
        self._tokenizer = tokenizer
        self.continuing_prefix = tokenizer.decoder.prefix if hasattr(tokenizer.decoder, 'prefix') else '##'
        self.s2c = Score2Chars(self.vocab, continuing_prefix=self.continuing_prefix, omit_eos=omit_eos, **(s2c_args or {}))
        self.pad_token_id = tokenizer.token_to_id(self.s2c.encode_single([self.pad_token]))
        self.eos_token_id = tokenizer.token_to_id(self.s2c.encode_single([self.eos_token]))
        self.sob_token_id = tokenizer.token_to_id(self.s2c.encode_single([self.sob_token]))
        if None in (self.pad_token_id, self.eos_token_id, self.sob_token_id):
            raise ValueError('the trained tokenizer lacks a single token for [PAD] / </s> / <bar>')

    @classmethod
    def from_file(cls, path: str, **kwargs):
        from tokenizers import Tokenizer
        tok = Tokenizer.from_file(path + '.json')
        with open(path + '_meta.json') as f:
            meta = json.load(f)
        ret = cls(tok, s2c_args=meta['score2chars'], **{**meta['music_vocab'], **kwargs})
        if meta.get('tok2id') and meta['tok2id'] != ret.vocab.tok2id:
            raise ValueError('the tokenizer file was trained on a different base vocabulary')
        return ret

    @property
    def vocab_size(self) -> int:
        return self._tokenizer.get_vocab_size()

    def __len__(self):
        return self.vocab_size

    def tokenize(self, text, mode: str = 'music') -> List[str]:
        toks = self._tokenizer.encode(self.s2c(text)).tokens
        return [self.s2c.decode_single(t) for t in toks] if mode == 'music' else toks

    def encode(self, text) -> List[int]:
        return self._tokenizer.encode(self.s2c(text)).ids

    def _convert_id_to_token(self, index: int) -> str:
        return self.s2c.decode_single(self._tokenizer.id_to_token(int(index)))

    def convert_ids_to_tokens(self, ids):
        return self._convert_id_to_token(ids) if isinstance(ids, int) else [self._convert_id_to_token(i) for i in ids]

    def convert_tokens_to_ids(self, toks):
        one = isinstance(toks, str)
        out = [self._tokenizer.token_to_id(self.s2c.encode_single(t)) for t in ([toks] if one else toks)]
        return out[0] if one else out

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        ids = ids.tolist() if hasattr(ids, 'tolist') else list(ids)
        if skip_special_tokens:
            ids = [i for i in ids if i != self.pad_token_id]
        return ' '.join(self._convert_id_to_token(i) for i in ids)

    def id2base_ids(self) -> List[List[int]]:
        return [[self.vocab.t2i(t) for t in self._convert_id_to_token(i).split()] for i in range(self.vocab_size)]
