"""CPU checks of the evaluation-metric restatement (oracle/metrics_ref.py) and of the host-side tables of
symbolic_music_generation_amd.metrics (no GPU, no compute calls into the library)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import metrics_ref as R  # noqa: E402


def test_key_tables_follow_the_reference():
    # elm_type.py: 24 keys, majors first; offsets of the tonic; E minor carries the reference's 'E-' entry (offset 3)
    assert len(R.KEY_STRS) == 24 and R.KEY_STRS[0] == 'CMajor' and R.KEY_STRS[12] == 'AMinor' and R.KEY_STRS[23] == 'EMinor'
    assert R.key_type_offset(R.KEY_STRS.index('CMajor')) == (1, 0)
    assert R.key_type_offset(R.KEY_STRS.index('GMajor')) == (1, 7)
    assert R.key_type_offset(R.KEY_STRS.index('AMinor')) == (0, 9)
    assert R.key_type_offset(R.KEY_STRS.index('EbMinor')) == (0, 3)
    assert R.key_type_offset(R.KEY_STRS.index('EMinor')) == (0, 3)


def test_in_key_ratio_hand_cases():
    c_major_scale = [60, 62, 64, 65, 67, 69, 71]                  # C D E F G A B
    assert R.in_key_ratio(c_major_scale, R.KEY_STRS.index('CMajor')) == 1.0
    assert R.in_key_ratio([61, 63, 66, 68, 70], R.KEY_STRS.index('CMajor')) == 0.0      # the five black keys
    assert R.in_key_ratio(c_major_scale + [61], R.KEY_STRS.index('CMajor')) == pytest.approx(7 / 8)
    # G major: F natural is off-key, F# is in
    assert R.in_key_ratio([65], R.KEY_STRS.index('GMajor')) == 0.0 and R.in_key_ratio([66], R.KEY_STRS.index('GMajor')) == 1.0
    # A minor (harmonic table [1,4,6,9,11] off): A B C D E F in, G and G# both in (offsets 10, 11 -> 11 is listed off)
    am = R.KEY_STRS.index('AMinor')
    assert R.in_key_ratio([69, 71, 72, 74, 76, 77, 79], am) == 1.0 and R.in_key_ratio([68], am) == 0.0
    assert R.in_key_ratio([], am) == 0.0                               # no pitch: counted as all off-key


def test_ids2pitches_and_metrics_on_token_streams():
    toks = ['<bar>', 'p_1/4', 'd_1', 'p_r', 'd_1/2', 'p_rare', 'p_8/3_5', 'TimeSig_4/4', 'p_12/-1']
    assert R.ids2pitches(toks) == [60, 55, 11]
    id2tok = lambda i: toks[i]
    preds = np.array([[1, 3, 6, 8, 0, 2]])
    labels = np.array([[5, 5, 5, -100, 5, 5]])
    ks = np.zeros((1, 24)); ks[0, 0] = 0.75; ks[0, 12] = 0.25
    # counted positions 0,1,2,4,5 -> tokens p_1/4, p_r, p_8/3_5, <bar>, d_1 -> pitches [60, 55]: both in C major and A minor
    assert R.ikr(preds, labels, id2tok, key_scores=ks, mode='vanilla') == pytest.approx(1.0)
    ks2 = np.zeros((1, 24)); ks2[0, R.KEY_STRS.index('BMajor')] = 1.0        # B major: C (1) off, G (8) off
    assert R.ikr(preds, labels, id2tok, key_scores=ks2, mode='vanilla') == pytest.approx(0.0)
    # next-token accuracy: pairs (pred[j], label[j+1]) with label[j+1] != -100
    p = np.array([[5, 9, 7, 5, 1, 0]]); lab = np.array([[0, 5, 5, -100, 5, 1]])
    assert R.ntp_acc(p, lab) == pytest.approx(3 / 4)                 # j=0 hit, j=1 miss, j=3 hit, j=4 hit; j=2 ignored
    assert R.ntp_acc(p[:, :-1], lab, clm_pred_shifted=True) == pytest.approx(3 / 4)


def test_host_tables_match_the_oracle():
    from symbolic_music_generation_amd import metrics as M
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    tab = M.in_key_table()
    for o in range(24):
        for pc in range(12):
            assert tab[o, pc] == R.in_key_ratio([60 + pc], o)
    for kind in ('midi', 'degree'):
        tok = MusicTokenizer(pitch_kind=kind)
        pcs = M.pitch_class_table(tok.vocab)
        for i in range(len(tok.vocab)):
            got = R.ids2pitches([tok.vocab.i2t(i)])
            assert (pcs[i] == -1) if not got else (pcs[i] == got[0] % 12)
        assert pcs[tok.vocab.t2i('p_r')] == -1 and pcs[tok.vocab.t2i('p_rare')] == -1
