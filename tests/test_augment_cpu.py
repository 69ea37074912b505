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


def test_tempo_group_oracle_hand_cases():
    f = lambda n, b=5: R.tempo_group(['TimeSig_4/4', f'Tempo_{n}', '<bar>'], b)[1]
    assert f(40) == 'Tempo_40/44' and f(44) == 'Tempo_40/44' and f(45) == 'Tempo_45/49' and f(120) == 'Tempo_120/124'
    assert f(234) == 'Tempo_230/234' and f(235) == 'Tempo_235/240' and f(240) == 'Tempo_235/240'      # last bin: one longer
    assert f(59, 20) == 'Tempo_40/59' and f(240, 20) == 'Tempo_220/240'
    assert R.tempo_group(['TimeSig_4/4', 'Tempo_low', '<bar>'])[1] == 'Tempo_low'


@pytest.mark.parametrize('pitch_shift', [False, True])
def test_tempo_group_table_matches_string_transform(env, pitch_shift):
    """TempoGroup as a table (alone, and folded into the pitch-shift tables) == the string transforms in dataset.py's order
    (crop -> tempo group -> key insert -> pitch shift), tokenised by the grouped-tempo vocabulary; every tempo is covered."""
    from symbolic_music_generation_amd.data import Augment, tempo_group_table
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    ts, td, ids, toks = env
    tg = MusicTokenizer(pitch_kind='degree' if pitch_shift else 'step', tempo_bin=5)
    assert len(tg.vocab) == len((td if pitch_shift else ts).vocab) - 201 + 40
    aug = Augment(ts, random_crop=True, insert_key=pitch_shift, keys=['AMajor'], pitch_shift=pitch_shift, seed=2,
                  group_tempo=True, **(dict(tokenizer_degree=tg) if pitch_shift else dict(tokenizer_group=tg)))
    parts, ordinal = aug.pieces(0, ids, crop_idx=7, key='AMajor' if pitch_shift else None)
    got = aug.tables[ordinal][np.concatenate(parts)]
    want = R.tempo_group(R.random_crop(toks, 7))
    if pitch_shift:
        want = R.pitch_shift(R.key_insert(want, 'AMajor'))
    assert want[1] != toks[1] and '/' in want[1]
    assert [tg.vocab.i2t(int(i)) for i in got] == want
    if not pitch_shift:
        tab = tempo_group_table(ts.vocab, tg.vocab)[0]
        for n in range(40, 241):
            assert tg.vocab.i2t(int(tab[ts.vocab.t2i(f'Tempo_{n}')])) == R.tempo_group(['x', f'Tempo_{n}'])[1]
        for t in ('Tempo_low', 'Tempo_high', '<bar>', 'd_1', 'p_r', '[OMIT]', 'TimeSig_4/4'):
            assert tg.vocab.i2t(int(tab[ts.vocab.t2i(t)])) == t
    with pytest.raises(ValueError):
        Augment(ts, group_tempo=True, tokenizer_group=ts)


@pytest.mark.parametrize('mode', ['full', 'swap'])
def test_channel_mixup_on_ids_matches_string_transform(env, mode):
    """ChannelMixer (transform.py:331-450) on ids == on token strings, with the same sequence of random draws"""
    from symbolic_music_generation_amd.data import Augment, channel_mix_ids
    ts, td, ids, toks = env
    assert toks.count('<tup>') > 0 and toks.count('<bass>') > 10
    aug = Augment(ts, channel_mixup=mode, seed=11)
    got = channel_mix_ids(ids, aug.mix_ids, mode, np.random.default_rng(3))
    r = np.random.default_rng(3)
    want = R.channel_mix(toks, mode, rand=lambda: float(r.random()), coin=lambda: int(r.integers(2)) == 0)
    assert [ts.vocab.i2t(int(i)) for i in got] == want
    # a permutation inside every bar that keeps each channel's own order: same multiset of notes, same bar count
    strip = lambda seq: sorted(t for t in seq if t not in ('<melody>', '<bass>'))
    assert strip(want) == strip(toks) and want.count('<bar>') == toks.count('<bar>') and want[-1] == '</s>'
    assert want != toks
    head, bars0 = R.split_elements(toks)
    _, bars1 = R.split_elements(want)
    for b0, b1 in zip(bars0, bars1):
        def chan(bar):
            out, cur = {'<melody>': [], '<bass>': []}, None
            for e in bar:
                if e[0] in out:
                    cur = e[0]
                else:
                    out[cur].append(e)
            return out
        assert chan(b0) == chan(b1)
    # through Augment.pieces: crop + key insertion first, mix-up last (dataset.py:331-350)
    aug2 = Augment(ts, random_crop=True, insert_key=True, keys=['GMajor'], channel_mixup=mode, seed=5)
    parts, _ = aug2.pieces(0, ids, crop_idx=4, key='GMajor')
    mixed = [ts.vocab.i2t(int(i)) for i in np.concatenate(parts)]
    base = R.key_insert(R.random_crop(toks, 4), 'GMajor')
    assert mixed[:4] == base[:4] and strip(mixed) == strip(base)


def test_channel_mixup_hand_cases():
    song = ('TimeSig_4/4 Tempo_120 <bar> <melody> p_1/4 d_1 <tup> p_2/4 p_3/4 p_4/4 d_2 </tup> <bass> p_5/2 d_4 '
            '<bar> <bass> p_6/2 d_4 </s>').split()
    draws = iter([0.9, 0.1, 0.1])              # thresh 2/3: bass, melody, (melody exhausted? no) ...
    out = R.channel_mix(song, 'full', rand=lambda: next(draws), coin=None)
    assert out[:2] == ['TimeSig_4/4', 'Tempo_120']
    # bar 1: bass first (0.9 >= 2/3) -> bass exhausted -> remaining melody with its marker; bar 2: bass only -> the reference
    # emits the notes WITHOUT a marker (add_to_melody is None)
    assert out[2:] == ['<bar>', '<bass>', 'p_5/2', 'd_4', '<melody>', 'p_1/4', 'd_1', '<tup>', 'p_2/4', 'p_3/4', 'p_4/4', 'd_2',
                       '</tup>', '<bar>', 'p_6/2', 'd_4', '</s>']
    sw = R.channel_mix(song, 'swap', rand=None, coin=lambda: False)
    assert sw[2:9] == ['<bar>', '<bass>', 'p_5/2', 'd_4', '<melody>', 'p_1/4', 'd_1']
