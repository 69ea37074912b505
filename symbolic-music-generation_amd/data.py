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
"""
import json
import os
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import lib, check, MusicXLError


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


class DeviceBatcher:
    """Double-buffered: while the model works on batch k, batch k+1 is being gathered into the other pinned buffer and copied."""

    def __init__(self, tf: TokenFile, batch_size: int, max_length: int, pad_id: int, device, shuffle: bool = False,
                 seed: int = 0, rank: int = 0, world: int = 1, drop_last: bool = False):
        if not torch.cuda.is_available():
            raise MusicXLError('DeviceBatcher needs a GPU (the pad/label step is a device kernel; no CPU fallback)')
        self.tf, self.B, self.L, self.pad_id = tf, batch_size, max_length, pad_id
        self.dev = torch.device(device)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, seed, rank, world, drop_last
        self.epoch = 0
        tdt = torch.uint16 if tf.dtype == np.uint16 else torch.int32
        self._pin_tok = [torch.empty(batch_size * max_length, dtype=tdt).pin_memory() for _ in range(2)]
        self._pin_off = [torch.empty(batch_size + 1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._dev_tok = [torch.empty(batch_size * max_length, dtype=tdt, device=self.dev) for _ in range(2)]
        self._dev_off = [torch.empty(batch_size + 1, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=self.dev)
        self._done = [torch.cuda.Event(), torch.cuda.Event()]       # buffer consumed by the pack kernel
        self._first = [True, True]

    def order(self) -> np.ndarray:
        idx = np.arange(len(self.tf))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        return idx[self.rank::self.world]

    def __len__(self):
        n = len(self.order())
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def _stage(self, slot: int, rows: np.ndarray):
        """gather `rows` (truncated to max_length) into pinned slot `slot`, then enqueue the H2D copies on the side stream"""
        if not self._first[slot]:
            self._done[slot].synchronize()                       # the previous batch in this slot has been packed
        self._first[slot] = False
        tok = self._pin_tok[slot].numpy().view(self.tf.dtype)
        off = self._pin_off[slot].numpy()
        pos = 0
        off[0] = 0
        for j, i in enumerate(rows):
            a = self.tf[int(i)][:self.L]
            tok[pos:pos + a.size] = a
            pos += a.size
            off[j + 1] = pos
        with torch.cuda.stream(self._copy_stream):
            self._dev_tok[slot][:max(pos, 1)].copy_(self._pin_tok[slot][:max(pos, 1)], non_blocking=True)
            self._dev_off[slot][:len(rows) + 1].copy_(self._pin_off[slot][:len(rows) + 1], non_blocking=True)
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
                                           torch.cuda.current_stream(self.dev).cuda_stream), 'mxl_pack_clm_batch')
            self._done[slot].record(torch.cuda.current_stream(self.dev))
            yield ids, labels
            cur = nxt
