"""Flat pre-tokenised file format (host side only) and the CPU statement of the batch contract."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.data_ref import pad_and_label  # noqa: E402


def test_token_file_roundtrip(tmp_path):
    from symbolic_music_generation_amd.data import TokenFile, write_token_file
    rng = np.random.default_rng(0)
    seqs = [rng.integers(0, 1190, size=n) for n in (5, 0, 300, 1, 2048, 77)]
    n = write_token_file(str(tmp_path / 'corpus'), seqs, vocab_size=1190)
    tf = TokenFile(str(tmp_path / 'corpus'))
    assert n == len(tf) == 6 and tf.dtype == np.uint16 and list(tf.lengths()) == [5, 0, 300, 1, 2048, 77]
    for a, b in zip(seqs, (tf[i] for i in range(6))):
        assert np.array_equal(a, b)
    write_token_file(str(tmp_path / 'big'), [[70000, 3]], vocab_size=100000)          # wide vocabulary -> int32
    assert TokenFile(str(tmp_path / 'big')).dtype == np.int32 and list(TokenFile(str(tmp_path / 'big'))[0]) == [70000, 3]


def test_pad_and_label_contract():
    ids, labels = pad_and_label([[5, 6, 7], [], [1, 2, 3, 4, 9, 9]], max_length=4, pad_id=1)
    assert ids.tolist() == [[5, 6, 7, 1], [1, 1, 1, 1], [1, 2, 3, 4]]
    # every pad id becomes -100 in the labels -- including a genuine pad token inside a sequence (the HF collator's behaviour)
    assert labels.tolist() == [[5, 6, 7, -100], [-100] * 4, [-100, 2, 3, 4]]
