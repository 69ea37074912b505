"""Input pipeline -> device.

The reference tokenises strings inside `Dataset.__getitem__` on every access (musicnlp/preprocess/dataset.py:330-365) behind
four DataLoader workers, and its author notes the resulting low GPU utilisation (train.py:366).  Here a corpus is tokenised ONCE
into a flat token file; a batch is a handful of contiguous slices of that file copied into pinned memory, sent with one
asynchronous H2D copy on a side stream, and padded / truncated / labelled on the GPU (mxl_pack_clm_batch):

    write_token_file(path, sequences, vocab_size)   ->  path.tok (uint16 | int32 ids), path.idx (int64 offsets), path.json
    TokenFile(path)                                  ->  len(), [i] -> ids (memory-mapped, zero-copy)
    DeviceBatcher(tf, batch_size, max_length, pad_id, device, ...)  ->  iterator of (input_ids, labels) on the device,
                                                         the contract of dataset.py:361 + the collator at train.py:360

Under data parallelism rank r takes the strided shard r, r + world, ... of the (optionally shuffled) order.

On-the-fly augmentation happens on ids, not strings (`Augment`): RandomCrop at `<bar>` boundaries
(musicnlp/preprocess/transform.py:59-114) is a choice of slice, KeyInsert (:138-151) one inserted id, and the step -> degree
PitchShift (:154-237) an id -> id table per key applied by the pack kernel.
"""
import json
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import lib, check, MusicXLError
from .vocab import KEY_NAMES


def write_token_file(path: str, sequences: Iterable[Sequence[int]], vocab_size: int) -> int:
    dtype = np.uint16 if vocab_size <= 65536 else np.int32
    offs = [0]
    with open(path + '.tok', 'wb') as f:
        for seq in sequences:
            a = np.asarray(seq, dtype=np.int64)
            if a.size and (a.min() < 0 or a.max() >= vocab_size):
                raise ValueError('token id outside the vocabulary')
            f.write(a.astype(dtype).tobytes())
            offs.append(offs[-1] + a.size)
    np.asarray(offs, dtype=np.int64).tofile(path + '.idx')
    with open(path + '.json', 'w') as f:
        json.dump(dict(vocab_size=vocab_size, dtype=np.dtype(dtype).name, n_sequences=len(offs) - 1, n_tokens=offs[-1]), f)
    return len(offs) - 1


class TokenFile:
    def __init__(self, path: str):
        meta = json.load(open(path + '.json'))
        self.vocab_size, self.dtype = meta['vocab_size'], np.dtype(meta['dtype'])
        self.offsets = np.fromfile(path + '.idx', dtype=np.int64)
        n_tok = int(self.offsets[-1])
        self.tokens = np.memmap(path + '.tok', dtype=self.dtype, mode='r', shape=(n_tok,)) if n_tok else np.zeros(0, self.dtype)
        assert len(self.offsets) == meta['n_sequences'] + 1

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i: int) -> np.ndarray:
        return self.tokens[self.offsets[i]:self.offsets[i + 1]]

    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)


class MixedTokenFiles:
    """`ProportionMixingDataset` (musicnlp/preprocess/dataset.py:367-453) over token files: the concatenation of several
    datasets with every one larger than `k` capped at `k` sequences, the kept ones re-drawn by `sample()` (meant to be called
    once per epoch, dataset.py:422-431).  Index order as the reference: dataset 0's (sub-sampled) entries, then dataset 1's, ...
    Quacks like a `TokenFile` for `DeviceBatcher` (`len`, `[i]`, `dtype`, `vocab_size`); `flat_index(i)` is the position of
    entry i in the plain concatenation of the full datasets (what per-sequence side tables such as `Augment.keys` are indexed by).
    The draw is `torch.randperm(size)[:k]` as in the reference -- on torch's global generator when neither `generator` nor
    `seed` is given (the reference's behaviour), on a private generator re-seeded from (seed, epoch) by `sample(epoch)` when
    `seed` is: every data-parallel rank then draws the SAME sub-sample, which the rank-strided shards partition."""

    def __init__(self, files: Sequence[TokenFile], k: int, generator: Optional[torch.Generator] = None,
                 seed: Optional[int] = None):
        assert k is not None and len(files) > 0
        self.files, self.k, self.generator, self.seed = list(files), int(k), generator, seed
        if seed is not None and generator is None:
            self.generator = torch.Generator()
        if len({f.dtype for f in self.files}) != 1 or len({f.vocab_size for f in self.files}) != 1:
            raise ValueError('mixed token files must share one vocabulary and id width')
        self.dtype, self.vocab_size = self.files[0].dtype, self.files[0].vocab_size
        self.sizes = [min(len(f), self.k) for f in self.files]
        self._starts = np.concatenate([[0], np.cumsum(self.sizes)])
        self._full_starts = np.concatenate([[0], np.cumsum([len(f) for f in self.files])])
        self._sampled: list = [None] * len(self.files)
        self.sample()

    def sample(self, epoch: int = 0):
        if self.seed is not None:
            self.generator.manual_seed(int(self.seed) * 1000003 + int(epoch))
        for j, f in enumerate(self.files):
            if len(f) > self.k:
                self._sampled[j] = torch.randperm(len(f), generator=self.generator)[:self.k].numpy()

    def __len__(self):
        return int(self._starts[-1])

    def locate(self, i: int) -> Tuple[int, int]:
        """global index -> (dataset, index inside that dataset)"""
        if not 0 <= i < len(self):
            raise IndexError(i)
        j = int(np.searchsorted(self._starts, i, side='right')) - 1
        loc = i - int(self._starts[j])
        if self._sampled[j] is not None:
            loc = int(self._sampled[j][loc])
        return j, loc

    def flat_index(self, i: int) -> int:
        j, loc = self.locate(i)
        return int(self._full_starts[j]) + loc

    def __getitem__(self, i: int) -> np.ndarray:
        j, loc = self.locate(int(i))
        return self.files[j][loc]

    def lengths(self) -> np.ndarray:
        return np.asarray([len(self[i]) for i in range(len(self))], dtype=np.int64)


_T0 = dict(C=0, D=1, E=2, F=3, G=4, A=5, B=6)            # musicnlp/preprocess/key_finder.py:199-207


def tempo_group_map(vocab_group) -> dict:
    """{'Tempo_N': 'Tempo_s/e'} for every common tempo N: the bin of the grouped vocabulary that holds N (TempoGroup,
    transform.py:117-136, over `tempo_meta_map`, music_vocab.py:395-420; bins of `tempo_bin` tempi from 40, the last one a
    tempo longer).  The two rare tokens (Tempo_low / Tempo_high) exist in both vocabularies and keep their strings."""
    import re
    if not vocab_group.tempo_bin:
        raise ValueError('TempoGroup needs a vocabulary built with tempo_bin')
    out = {}
    for tok in vocab_group._tempos():
        s, e = map(int, re.match(r'^Tempo_(\d+)/(\d+)$', tok).groups())
        for n in range(s, e + 1):
            out[f'Tempo_{n}'] = tok
    return out


def vocab_translation_table(vocab_src, vocab_dst, tok_map: Optional[dict] = None) -> np.ndarray:
    """(V_src,) int32: id in `vocab_src` -> id of the same token string (or of tok_map[string]) in `vocab_dst`"""
    tok_map = tok_map or {}
    return np.asarray([vocab_dst.t2i(tok_map.get(vocab_src.i2t(i), vocab_src.i2t(i))) for i in range(len(vocab_src))], dtype=np.int32)


def tempo_group_table(vocab_none, vocab_group) -> np.ndarray:
    """(1, V_none) int32: TempoGroup as an id -> id table from the ungrouped-tempo vocabulary into the grouped one (the tempo
    token moves to its bin's token, every other token keeps its string; ids shift because the grouped vocabulary holds 40
    tempo tokens where the ungrouped one holds 201)."""
    return vocab_translation_table(vocab_none, vocab_group, tempo_group_map(vocab_group))[None]


def pitch_shift_tables(vocab_step, vocab_degree, tok_map: Optional[dict] = None) -> np.ndarray:
    """(24, V_step) int32: table[key ordinal][step-vocabulary id] = degree-vocabulary id.  Pitch tokens `p_i/o_S` become
    `p_i'/o'_deg` with deg = (t0[S] - t0[tonic letter]) % 7 + 1 (ScaleDegreeFinder.map_single, key_finder.py:245-262) and the
    two out-of-range step tokens folded back by an octave (transform.py:184-191); every other token keeps its string (or
    becomes tok_map[string]: the tempo grouping folded into the same table) and changes id space only."""
    import re
    tok_map = tok_map or {}
    pat = re.compile(r'^p_(-?\d+)/(-?\d+)_([A-G])$')
    V = len(vocab_step)
    tab = np.zeros((len(KEY_NAMES), V), dtype=np.int32)
    for o, key in enumerate(KEY_NAMES):
        k0 = _T0[key[0]]
        for i in range(V):
            tok = vocab_step.i2t(i)
            m = pat.match(tok)
            if m is None:
                tab[o, i] = vocab_degree.t2i(tok_map.get(tok, tok))
                continue
            midi = int(m.group(1)) - 1 + (int(m.group(2)) + 1) * 12
            midi = midi + 12 if midi == -12 else (midi - 12 if midi == 131 else midi)
            deg = (_T0[m.group(3)] - k0) % 7 + 1
            tab[o, i] = vocab_degree.t2i(f'p_{(midi % 12) + 1}/{midi // 12 - 1}_{deg}')
    return tab


def channel_mix_ids(seq: np.ndarray, ids: dict, mode: str, rng) -> np.ndarray:
    """`ChannelMixer` (musicnlp/preprocess/transform.py:331-450) on token ids: inside every bar the notes of the melody and the
    bass channel are re-interleaved at random, each channel keeping its own order and a channel marker emitted at every change
    (mode 'full': melody next with probability n_melody / (n_melody + n_bass), one draw per decision), or the two channel blocks
    are emitted in random order (mode 'swap').  A note is [pitch, duration], a tuplet everything from <tup> through </tup>.
    `ids`: token ids of <bar>, <melody>, <bass>, <tup>, </tup>, </s>, [OMIT] and the first / last key id.  Integer work on the
    host array while the batch is being gathered; `rng` is a numpy Generator (the draws `torch.rand` / `torch.randint` make in
    the reference)."""
    BAR, MEL, BASS, TUP, TUPE, EOS, OMIT = (ids[k] for k in ('bar', 'melody', 'bass', 'tup', 'tup_end', 'eos', 'omit'))
    seq = np.asarray(seq)
    n = len(seq)
    bars = np.flatnonzero(seq == BAR)
    if len(bars) == 0:
        return seq
    out = [seq[:bars[0]]]
    end = n - 1 if seq[n - 1] == EOS else n
    bounds = list(bars) + [end]
    marker = {True: np.asarray([MEL], dtype=seq.dtype), False: np.asarray([BASS], dtype=seq.dtype)}
    bar_tok = np.asarray([BAR], dtype=seq.dtype)
    for a, b in zip(bounds[:-1], bounds[1:]):
        i = a + 1
        if i >= b or seq[i] not in (MEL, BASS):
            raise MusicXLError('channel mix-up: every bar must open with a channel marker')
        mel, bass, cur = [], [], None
        while i < b:
            t = seq[i]
            if t == MEL:
                cur = mel; i += 1
            elif t == BASS:
                cur = bass; i += 1
            elif t == TUP:
                j = i + 1
                while seq[j] != TUPE:
                    j += 1
                cur.append(seq[i:j + 1]); i = j + 1
            else:
                cur.append(seq[i:i + 2]); i += 2
        out.append(bar_tok)
        if mode == 'full':
            n_m, n_b = len(mel), len(bass)
            thresh = n_m / (n_m + n_b)
            im = ib = 0
            prev, add_m = None, None
            while im < n_m and ib < n_b:
                add_m = bool(rng.random() < thresh)
                if add_m != prev:
                    out.append(marker[add_m])
                if add_m:
                    out.append(mel[im]); im += 1
                else:
                    out.append(bass[ib]); ib += 1
                prev = add_m
            if im < n_m:
                if not add_m:
                    out.append(marker[True])
                out += mel[im:]
            else:
                if add_m:                       # (the reference leaves a bass-only bar without its marker: add_m is None there)
                    out.append(marker[False])
                out += bass[ib:]
        else:
            tm, tb = [marker[True]] + mel, [marker[False]] + bass
            out += (tm + tb) if int(rng.integers(2)) == 0 else (tb + tm)
    out.append(np.asarray([EOS], dtype=seq.dtype))
    return np.concatenate(out)


class Augment:
    """Per-sequence augmentation on ids.  `keys[i]`: key name of sequence i ('CMajor', ...) or a {name: weight} dict to sample
    from (KeyInsert with `pt_sample`); needed for key insertion and pitch shift."""

    def __init__(self, tokenizer, random_crop: bool = False, min_seg_length: int = 16, crop_mult: int = 1,
                 insert_key: bool = False, keys=None, pitch_shift: bool = False, tokenizer_degree=None, seed: int = 0,
                 group_tempo: bool = False, tokenizer_group=None, channel_mixup=False):
        """`group_tempo`: the stored ids are in the ungrouped-tempo vocabulary of `tokenizer`; the batch comes out in the ids
        of the grouped one (`tokenizer_group`, or `tokenizer_degree` built with tempo_bin when pitch shift is on too --
        dataset.py:254-257,338-339 applies TempoGroup before KeyInsert / PitchShift; as tables they compose into one)."""
        v = tokenizer.vocab
        self.bar_id, self.omit_id = v.t2i(v.start_of_bar), v.t2i(v.omitted_segment)
        # channel mix-up (dataset.py:277-281, 348-350: the LAST transform, after key insertion / pitch shift; as those are
        # per-token id tables applied on the device, permuting the tokens first gives the same batch)
        self.channel_mixup = ('full' if channel_mixup is True else channel_mixup) or None
        if self.channel_mixup not in (None, 'full', 'swap'):
            raise ValueError(f'channel_mixup {channel_mixup!r}: full or swap')
        self.mix_ids = dict(bar=self.bar_id, omit=self.omit_id, melody=v.t2i('<melody>'), bass=v.t2i('<bass>'),
                            tup=v.t2i('<tup>'), tup_end=v.t2i('</tup>'), eos=v.t2i('</s>'))
        self.random_crop, self.min_seg_length, self.crop_mult = random_crop, min_seg_length, crop_mult
        self.insert_key, self.keys, self.pitch_shift = insert_key, keys, pitch_shift
        if (insert_key or pitch_shift) and keys is None:
            raise ValueError('key insertion / pitch shift need the key(s) of every sequence')
        if pitch_shift and not insert_key:
            raise ValueError('PitchShift reads the key token at position 2: enable insert_key (transform.py:219-221)')
        self.key_id = {k: v.t2i(f'Key_{k}') for k in KEY_NAMES}
        self.group_tempo = group_tempo
        if group_tempo:
            dst = (tokenizer_degree if pitch_shift else tokenizer_group)
            if dst is None or not dst.vocab.tempo_bin or v.tempo_bin:
                raise ValueError('group_tempo: `tokenizer` must be ungrouped and the target tokenizer built with tempo_bin')
            tmap = tempo_group_map(dst.vocab)
            self.tables = pitch_shift_tables(v, dst.vocab, tmap) if pitch_shift else tempo_group_table(v, dst.vocab)
        else:
            self.tables = pitch_shift_tables(v, tokenizer_degree.vocab) if pitch_shift else None
        self.rng = np.random.default_rng(seed)

    def crop_high(self, n_bar: int) -> int:
        if n_bar <= self.min_seg_length:
            return 0
        high = n_bar - self.min_seg_length
        if self.crop_mult == 1:
            return high
        return high // self.crop_mult if high >= self.crop_mult else 0

    def pieces(self, i: int, seq: np.ndarray, crop_idx: Optional[int] = None, key: Optional[str] = None):
        """-> (list of int arrays whose concatenation is the augmented sequence, key ordinal or -1).  `crop_idx` / `key`
        override the random draws (tests)."""
        parts = [seq]
        if self.random_crop:
            bars = np.flatnonzero(seq == self.bar_id)
            high = self.crop_high(len(bars))
            if crop_idx is None:
                crop_idx = int(self.rng.integers(0, high + 1)) * (1 if self.crop_mult == 1 else self.crop_mult) if high > 0 else 0
            if high > 0 and crop_idx != 0:
                parts = [seq[:bars[0]], np.asarray([self.omit_id], dtype=seq.dtype), seq[bars[crop_idx]:]]
        ordinal = -1
        if self.insert_key:
            if key is None:
                k = self.keys[i]
                if isinstance(k, dict):
                    names = list(k)
                    w = np.asarray([k[n] for n in names], dtype=np.float64)
                    k = names[int(self.rng.choice(len(names), p=w / w.sum()))]
                key = k
            ordinal = KEY_NAMES.index(key)
            head = parts[0]
            parts = [head[:2], np.asarray([self.key_id[key]], dtype=seq.dtype), head[2:]] + parts[1:]
        if self.channel_mixup:
            parts = [channel_mix_ids(np.concatenate(parts), self.mix_ids, self.channel_mixup, self.rng)]
        return parts, (ordinal if self.pitch_shift else (0 if self.group_tempo else -1))

class DeviceBatcher:
    """Double-buffered: while the model works on batch k, batch k+1 is being gathered into the other pinned buffer and copied."""

    def __init__(self, tf: TokenFile, batch_size: int, max_length: int, pad_id: int, device, shuffle: bool = False,
                 seed: int = 0, rank: int = 0, world: int = 1, drop_last: bool = False, augment: Optional[Augment] = None):
        if not torch.cuda.is_available():
            raise MusicXLError('DeviceBatcher needs a GPU (the pad/label step is a device kernel; no CPU fallback)')
        self.tf, self.B, self.L, self.pad_id = tf, batch_size, max_length, pad_id
        self.dev = torch.device(device)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.augment = augment
        self.epoch = 0
        tdt = torch.uint16 if tf.dtype == np.uint16 else torch.int32
        self._pin_tok = [torch.empty(batch_size * max_length, dtype=tdt).pin_memory() for _ in range(2)]
        self._pin_off = [torch.empty(batch_size + 1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._dev_tok = [torch.empty(batch_size * max_length, dtype=tdt, device=self.dev) for _ in range(2)]
        self._dev_off = [torch.empty(batch_size + 1, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self._pin_tab = [torch.empty(batch_size, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._dev_tab = [torch.empty(batch_size, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self._remap = None
        if augment is not None and augment.tables is not None:
            self._remap = torch.from_numpy(augment.tables).to(self.dev)
        self._copy_stream = torch.cuda.Stream(device=self.dev)
        self._done = [torch.cuda.Event(), torch.cuda.Event()]       # buffer consumed by the pack kernel
        self._first = [True, True]

    def order(self) -> np.ndarray:
        """This rank's rows.  Like `DistributedSampler` the (shuffled) index is padded by wrapping around to a multiple of the
        world size before striding: every rank then yields the same number of equally sized batches, so the ranks' gradient
        all-reduces always pair up (a shard one row short would run one step fewer and hang the job)."""
        idx = np.arange(len(self.tf))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        # evaluation (shuffle off) takes the plain strided shard: no duplicated rows in the all-reduced sums, and its only
        # collective is one sum at the end, so ranks may differ by one batch
        if self.shuffle and self.world > 1 and len(idx) % self.world:
            idx = np.concatenate([idx, idx[:self.world - len(idx) % self.world]])
        return idx[self.rank::self.world]

    def __len__(self):
        n = len(self.order())
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def n_rows(self) -> int:
        """sequences in the whole dataset (all ranks): what steps-per-epoch is computed from"""
        return len(self.tf)

    def _stage(self, slot: int, rows: np.ndarray):
        """gather `rows` (truncated to max_length) into pinned slot `slot`, then enqueue the H2D copies on the side stream"""
        if not self._first[slot]:
            self._done[slot].synchronize()                       # the previous batch in this slot has been packed
        self._first[slot] = False
        tok = self._pin_tok[slot].numpy().view(self.tf.dtype)
        off = self._pin_off[slot].numpy()
        pos = 0
        off[0] = 0
        tab = self._pin_tab[slot].numpy()
        for j, i in enumerate(rows):
            if self.augment is None:
                parts, tab[j] = [self.tf[int(i)]], -1
            else:
                src = self.tf.flat_index(int(i)) if hasattr(self.tf, 'flat_index') else int(i)
                parts, tab[j] = self.augment.pieces(src, self.tf[int(i)])
            room = self.L
            for a in parts:                                   # truncation=True: the first max_length tokens of the result
                a = a[:room]
                tok[pos:pos + a.size] = a
                pos += a.size
                room -= a.size
                if room == 0:
                    break
            off[j + 1] = pos
        with torch.cuda.stream(self._copy_stream):
            self._dev_tok[slot][:max(pos, 1)].copy_(self._pin_tok[slot][:max(pos, 1)], non_blocking=True)
            self._dev_off[slot][:len(rows) + 1].copy_(self._pin_off[slot][:len(rows) + 1], non_blocking=True)
            self._dev_tab[slot][:len(rows)].copy_(self._pin_tab[slot][:len(rows)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return ev, len(rows)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        order = self.order()
        nb = len(self)
        self.epoch += 1
        if nb == 0:
            return
        cur = self._stage(0, order[:self.B])
        for k in range(nb):
            slot = k & 1
            nxt = self._stage(slot ^ 1, order[(k + 1) * self.B:(k + 2) * self.B]) if k + 1 < nb else None
            ev, nrow = cur
            torch.cuda.current_stream(self.dev).wait_event(ev)
            ids = torch.empty(nrow, self.L, device=self.dev, dtype=torch.int64)
            labels = torch.empty_like(ids)
            check(lib().mxl_pack_clm_batch(self._dev_tok[slot].data_ptr(), self.tf.dtype.itemsize, self._dev_off[slot].data_ptr(),
                                           ids.data_ptr(), labels.data_ptr(), nrow, self.L, self.pad_id,
                                           self._remap.data_ptr() if self._remap is not None else None,
                                           self._dev_tab[slot].data_ptr() if self._remap is not None else None,
                                           self._remap.shape[1] if self._remap is not None else 0,
                                           torch.cuda.current_stream(self.dev).cuda_stream), 'mxl_pack_clm_batch')
            self._done[slot].record(torch.cuda.current_stream(self.dev))
            yield ids, labels
            cur = nxt
