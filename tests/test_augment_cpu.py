"""Augmentation on ids (data.Augment, pitch-shift tables) against the string-level restatement of the reference transforms
(oracle/augment_ref.py), on the reference's real step-pitch token stream (tests/golden/sample_score_ids.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import augment_ref as R  # noqa: E402


@pytest.fixture(scope='module')
def env():
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    ts, td = MusicTokenizer(pitch_kind='step'), MusicTokenizer(pitch_kind='degree')
    ids = np.load(os.path.join(ROOT, 'tests', 'golden', 'sample_score_ids.npz'))['sample_full_step'].astype(np.int64)
    toks = [ts.vocab.i2t(int(i)) for i in ids]
    assert toks[0].startswith('TimeSig_') and toks[1].startswith('Tempo_') and toks.count('<bar>') > 40
    return ts, td, ids, toks


def test_oracle_hand_cases():
    song = ['TimeSig_4/4', 'Tempo_120'] + sum([['<bar>', f'p_{i % 12 + 1}/4_C', 'd_1'] for i in range(20)], [])
    assert R.crop_high(song, 16, 1) == 4 and R.crop_high(song, 16, 4) == 1 and R.crop_high(song[:2 + 3 * 16], 16) == 0
    c = R.random_crop(song, 3)
    assert c[:3] == ['TimeSig_4/4', 'Tempo_120', '[OMIT]'] and c[3:] == song[2 + 3 * 3:] and c.count('<bar>') == 17
    assert R.random_crop(song, 0) == song
    k = R.key_insert(song, 'GMajor')
    assert k[2] == 'Key_GMajor' and len(k) == len(song) + 1
    # G major: G -> 1, A -> 2, F -> 7 ; midi from index/octave ; rests and non-pitch tokens untouched
    s = ['TimeSig_4/4', 'Tempo_120', 'Key_GMajor', 'p_8/4_G', 'p_10/4_A', 'p_6/4_F', 'p_r', 'd_1', 'p_1/-2_B', 'p_12/9_C']
    assert R.pitch_shift(s) == ['TimeSig_4/4', 'Tempo_120', 'Key_GMajor', 'p_8/4_1', 'p_10/4_2', 'p_6/4_7', 'p_r', 'd_1',
                                'p_1/-1_3', 'p_12/8_4']


@pytest.mark.parametrize('crop_idx,key', [(0, 'CMajor'), (5, 'GMajor'), (17, 'EbMinor'), (None, 'F#Minor')])
def test_augment_on_ids_matches_string_transforms(env, crop_idx, key):
    from symbolic_music_generation_amd.data import Augment
    ts, td, ids, toks = env
    aug = Augment(ts, random_crop=True, insert_key=True, keys=[key], pitch_shift=True, tokenizer_degree=td, seed=4)
    n_bar = toks.count('<bar>')
    if crop_idx is None:
        crop_idx = aug.crop_high(n_bar)                    # the last admissible crop point
    assert aug.crop_high(n_bar) == R.crop_high(toks)
    parts, ordinal = aug.pieces(0, ids, crop_idx=crop_idx, key=key)
    got_step = np.concatenate(parts)
    want = R.pitch_shift(R.key_insert(R.random_crop(toks, crop_idx), key))
    # id-space result: step ids through the key's table == degree ids of the string result
    got = aug.tables[ordinal][got_step]
    assert [td.vocab.i2t(int(i)) for i in got] == want
    assert got.tolist() == [td.vocab.t2i(t) for t in want]


def test_random_draws_stay_in_range(env):
    from symbolic_music_generation_amd.data import Augment
    ts, td, ids, toks = env
    aug = Augment(ts, random_crop=True, insert_key=True, keys=[dict(CMajor=0.7, AMinor=0.3)], seed=1)
    seen_keys, seen_lens = set(), set()
    for _ in range(50):
        parts, ordinal = aug.pieces(0, ids)
        out = np.concatenate(parts)
        seen_keys.add(ts.vocab.i2t(int(out[2]))); seen_lens.add(len(out))
        assert ordinal == -1 and (out == aug.bar_id).sum() >= 16
    assert seen_keys == {'Key_CMajor', 'Key_AMinor'} and len(seen_lens) > 5
