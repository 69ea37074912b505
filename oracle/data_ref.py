"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's batch contract:
`tokenizer(toks, padding='max_length', truncation=True)` (musicnlp/preprocess/dataset.py:361; MusicTokenizer pads on the right
with `[PAD]`, music_tokenizer.py:28-40) followed by DataCollatorForLanguageModeling(mlm=False) (musicnlp/trainer/train.py:360):
labels = input_ids.clone(); labels[labels == pad_token_id] = -100.  Also `ProportionMixingDataset` (dataset.py:367-453) and the
bar-aligned cuts around generation (musicnlp/trainer/eval.py:178-198).  Parity unpinned: `musicnlp` cannot be imported here
(stefutil / music21 missing) and the reference holds no tests for these functions; each restatement follows the cited lines.
Only tests/ may import this module."""
import numpy as np


def pad_and_label(seqs, max_length: int, pad_id: int):
    B = len(seqs)
    ids = np.full((B, max_length), pad_id, dtype=np.int64)
    for b, s in enumerate(seqs):
        s = np.asarray(s, dtype=np.int64)[:max_length]           # truncation=True
        ids[b, :len(s)] = s                                      # padding='max_length', right side
    labels = ids.copy()
    labels[labels == pad_id] = -100
    return ids, labels


class ProportionMixingRef:
    """`ProportionMixingDataset` (musicnlp/preprocess/dataset.py:367-453) on plain lists: datasets larger than k are cut to k
    entries chosen by `torch.randperm(size)[:k]` (global generator), re-drawn by `sample()`; a global index walks the datasets in
    order (`_idx2dset_idx`, :437-445) and goes through the sub-sample index when there is one (:447-455)."""

    def __init__(self, dataset_list, k: int):
        self.dsets, self.k = dataset_list, k
        self.dset_szs = [min(len(d), k) for d in self.dsets]
        self.sz = sum(self.dset_szs)
        self._sampled_idxs = [None] * len(self.dsets)
        self.sample()

    def sample(self):
        import torch
        for i, dset in enumerate(self.dsets):
            sz = len(dset)
            if sz > self.k:
                self._sampled_idxs[i] = torch.randperm(sz)[:self.k]

    def __len__(self):
        return self.sz

    def __getitem__(self, idx: int):
        for i, sz in enumerate(self.dset_szs):
            if idx < sz:
                if self._sampled_idxs[i] is not None:
                    idx = self._sampled_idxs[i][idx].item()
                return self.dsets[i][idx]
            idx -= sz
        raise ValueError('Should not happen')


def truncate_last_bar_ref(ids, sob_id: int):
    """MusicGenerator._truncate_last_bar (musicnlp/trainer/eval.py:178-185) on a list of ints"""
    idxs = [i for i, t in enumerate(ids) if t == sob_id]
    assert len(idxs) > 0
    return list(ids[:idxs[-1]])


def truncate_first_n_bar_ref(ids, sob_id: int, n_bar: int = 8):
    """MusicGenerator.truncate_first_n_bar (eval.py:187-198) on ids: prefix up to the n_bar-th start-of-bar + a start-of-bar"""
    idxs = [i for i, t in enumerate(ids) if t == sob_id]
    return list(ids[:idxs[n_bar]]) + [sob_id]
