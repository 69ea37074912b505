"""Sub-word tokenizers (subword.py; reference musicnlp/trainer/pair_merge_tokenizer.py, wordpiece_tokenizer.py) trained and
exercised on the reference's real token streams (tests/golden/sample_score_ids.npz, from musicnlp/_sample_score.py)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def songs():
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'sample_score_ids.npz'))
    out = {}
    for kind in ('midi', 'step', 'degree'):
        tok = MusicTokenizer(pitch_kind=kind)
        out[kind] = ' '.join(tok.vocab.i2t(int(i)) for i in z[f'sample_full_{kind}'])
    return out


def test_split_song_groups_elements(songs):
    from symbolic_music_generation_amd.subword import split_song
    from symbolic_music_generation_amd.vocab import MusicVocabulary
    v = MusicVocabulary(pitch_kind='degree')
    s = split_song(v, songs['degree'])
    assert s.time_sig.startswith('TimeSig_') and s.tempo.startswith('Tempo_') and s.key.startswith('Key_') and s.end_of_song == '</s>'
    flat = [s.time_sig, s.tempo, s.key] + sum((['<bar>'] + sum(b, []) for b in s.elms_by_bar), []) + ['</s>']
    assert flat == songs['degree'].split()
    for bar in s.elms_by_bar:
        assert bar[0] in (['<melody>'], ['<bass>'])
        for e in bar:
            assert len(e) == 1 or (len(e) == 2 and e[0].startswith('p_') and e[1].startswith('d_')) or \
                (e[0] == '<tup>' and e[-1] == '</tup>' and e[-2].startswith('d_'))


def tr_counts(tr, song):
    from collections import Counter
    return Counter(tr.song2elements(song))


@pytest.mark.parametrize('kind', ['midi', 'degree'])
def test_pair_merge_tokenizer(tmp_path, songs, kind):
    from symbolic_music_generation_amd.subword import PairMergeTokenizer, PairMergeTokenizerTrainer
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    song = songs[kind]
    tr = PairMergeTokenizerTrainer(pitch_kind=kind)
    base = len(tr.vocab)
    meta = tr([song], coverage_ratio=0.8, save=str(tmp_path / 'pm'))
    elms = tr.song2elements(song)
    assert meta['n_unique'] == len(set(elms)) and meta['original_vocab_size'] == base
    added = meta['added_tok2id']
    assert 0 < len(added) < meta['n_unique'] and sorted(added.values()) == list(range(base, base + len(added)))
    counts = [meta['occurence_count'][t] for t in sorted(added, key=added.get)]
    assert counts == sorted(counts, reverse=True)                              # ids in descending order of frequency
    covered = sum(counts) / len(elms)
    nxt = sorted(tr_counts(tr, song).values(), reverse=True)[len(counts)]
    assert covered <= 0.8 < covered + nxt / len(elms)                          # np.searchsorted(ratio, r, side='right') of the reference
    tok = PairMergeTokenizer.from_file(str(tmp_path / 'pm'))
    assert tok.vocab_size == base + len(added) and tok.pitch_kind == kind
    ids = tok.encode(song)
    toks = tok.tokenize(song)
    assert len(ids) < len(song.split()) and tok.decode(ids) == song            # merged elements shorten it; lossless
    assert any(i >= base for i in ids) and all(' ' in t for t, i in zip(toks, ids) if i >= base)
    vt = MusicTokenizer(pitch_kind=kind)
    expanded = [b for i in ids for b in tok.id2base_ids()[i]]
    assert expanded == vt.encode(song)
    enc = tok(song, padding='max_length', truncation=True, max_length=64)
    assert len(enc['input_ids']) == 64 and list(enc.keys()) == ['input_ids']
    with pytest.raises(ValueError):
        tr([song], vocab_size=2000, coverage_ratio=0.5)
    full = tr([song], vocab_size=base + 10 ** 6)                               # more than there are elements: all added
    assert full['n_added'] == full['n_unique'] and full['coverage_ratio'] == 1.0


def test_wordpiece_tokenizer(tmp_path, songs):
    pytest.importorskip('tokenizers')
    from symbolic_music_generation_amd.subword import Score2Chars, WordPieceMusicTokenizer, WordPieceMusicTokenizerTrainer
    from symbolic_music_generation_amd.vocab import MusicTokenizer, MusicVocabulary
    song = songs['midi']
    v = MusicVocabulary(pitch_kind='midi', is_wordpiece=True)
    s2c = Score2Chars(v, independent_global_token=True, punctuate=True)
    chars = s2c(song)
    assert s2c.decode(chars) == song and len(set(s2c.dec_chars)) == len(v) and ' ' not in s2c.dec_chars
    words = s2c.split(song)
    assert words[0] == [song.split()[0]] and ['<bar>'] in words and all(len(w) == 1 for w in words if w[0] in s2c.spec_toks)
    tr = WordPieceMusicTokenizerTrainer(pitch_kind='midi')
    tok = tr([song] * 4, vocab_size=len(v) + 150, save=str(tmp_path / 'wp'))
    assert len(v) < tok.vocab_size <= len(v) + 150
    ids = tok.encode(song)
    assert len(ids) < len(song.split()) and tok.decode(ids) == song
    base = MusicTokenizer(pitch_kind='midi')
    assert [b for i in ids for b in tok.id2base_ids()[i]] == base.encode(song)
    assert tok.convert_ids_to_tokens(tok.pad_token_id) == '[PAD]' and tok.convert_ids_to_tokens(tok.eos_token_id) == '</s>'
    tok2 = WordPieceMusicTokenizer.from_file(str(tmp_path / 'wp'))
    assert tok2.encode(song) == ids and tok2.vocab_size == tok.vocab_size
    enc = tok2([song, song], padding='max_length', truncation=True, max_length=100, return_tensors='pt')
    assert tuple(enc['input_ids'].shape) == (2, 100)


def test_factory_accepts_subword_schemes_and_vocab_size_sets_cutoffs(tmp_path, songs):
    """train.py:31-59 with tokenize_scheme = pairmerge: the tokenizer's vocabulary size picks the adaptive-softmax cutoffs
    (models/transformer_xl.py:53-66)"""
    from symbolic_music_generation_amd.subword import PairMergeTokenizer, PairMergeTokenizerTrainer
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
    from symbolic_music_generation_amd.trainer import get_model_n_tokenizer
    tr = PairMergeTokenizerTrainer(pitch_kind='midi')
    tr([songs['midi']], coverage_ratio=0.9, save=str(tmp_path / 'pm'))
    tok = PairMergeTokenizer.from_file(str(tmp_path / 'pm'))
    assert tok.vocab_size < 1000
    assert MyTransfoXLConfig('debug', tokenizer=tok).cutoffs == [] and MyTransfoXLConfig('debug', tokenizer=tok).vocab_size == tok.vocab_size
    tr2 = PairMergeTokenizerTrainer(pitch_kind='degree')
    meta = tr2([songs['degree']], vocab_size=len(tr2.vocab) + 100, save=str(tmp_path / 'pmd'))
    tok2 = PairMergeTokenizer.from_file(str(tmp_path / 'pmd'))
    assert tok2.vocab_size == 1190 + 100 and MyTransfoXLConfig('debug', tokenizer=tok2).cutoffs == [1000]
    with pytest.raises(ValueError):
        get_model_n_tokenizer('transf-xl', 'debug', tokenize_scheme='pairmerge')          # needs a tokenizer file
    with pytest.raises(ValueError):
        get_model_n_tokenizer('transf-xl', 'debug', tokenize_scheme='bpe')


# ---------------------------------------------------------------------------------------------------------------------------
# against the oracle (oracle/subword_ref.py: the reference's algorithms restated on plain token strings)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kind', ['midi', 'step', 'degree'])
def test_pair_merge_matches_oracle(tmp_path, songs, kind):
    from oracle import subword_ref as R
    from symbolic_music_generation_amd.subword import PairMergeTokenizer, PairMergeTokenizerTrainer, split_song
    song = songs[kind]
    toks = song.split()
    tr = PairMergeTokenizerTrainer(pitch_kind=kind)
    base = len(tr.vocab)
    s, o = split_song(tr.vocab, song), R.str2tok_elms(toks)
    assert (s.time_sig, s.tempo, s.key, s.omit, s.end_of_song) == (o['time_sig'], o['tempo'], o['key'], o['omit'], o['end_of_song'])
    assert s.elms_by_bar == o['elms_by_bar']
    assert tr.song2elements(song) == R.song2uniq_elms(toks)
    for kw in (dict(coverage_ratio=0.8), dict(coverage_ratio=0.35), dict(vocab_size=base + 40), dict(vocab_size=base + 10 ** 6)):
        meta = tr([song], save=str(tmp_path / 'pm'), **kw)
        want = R.pair_merge_train([toks], base, **kw)
        assert meta['added_tok2id'] == want['added_tok2id'] and meta['n_unique'] == want['n_unique']
        assert meta['n_added'] == want['n_added'] and meta['occurence_count'] == want['occurence_count']
        tok = PairMergeTokenizer.from_file(str(tmp_path / 'pm'))
        assert tok.tokenize(song) == R.pair_merge_tokenize(toks, want['added_tok2id'])
    with pytest.raises(ValueError):
        R.pair_merge_train([toks], base, vocab_size=base + 10, coverage_ratio=0.5)


def test_wordpiece_pre_and_post_processing_match_oracle(tmp_path, songs):
    pytest.importorskip('tokenizers')
    from oracle import subword_ref as R
    from symbolic_music_generation_amd.subword import Score2Chars, WordPieceMusicTokenizerTrainer, _uni_chars
    from symbolic_music_generation_amd.vocab import MusicVocabulary
    assert _uni_chars() == R.uni_chars()
    for kind in ('midi', 'degree'):
        song = songs[kind]
        toks = song.split()
        v = MusicVocabulary(pitch_kind=kind, is_wordpiece=True)
        # (True, True) is the setting the reference trains and loads with (wordpiece_tokenizer.py:606,650); (False, False) is
        # the unsplit form
        for indep, punct in ((True, True), (False, False)):
            s2c = Score2Chars(v, independent_global_token=indep, punctuate=punct)
            words = R.score2words(toks, indep, punct)
            assert s2c.split(song) == words
            assert s2c(song) == R.words2chars(words, v.t2i, R.uni_chars()[:len(v)])
        # the reference's punctuate-only branch (never used by it) builds its first word from the time signature and tempo alone
        # and DROPS a key / [OMIT] token (wordpiece_tokenizer.py:166-170); the product keeps them in that word so that the
        # encoding stays lossless.  Everything after the first word is identical.
        s2c = Score2Chars(v, independent_global_token=False, punctuate=True)
        ref_words, got = R.score2words(toks, False, True), s2c.split(song)
        assert got[1:] == ref_words[1:] and got[0][:2] == ref_words[0] and s2c.decode(s2c(song)) == song
    # a trained model: its ids re-derived from its vocabulary alone by greedy longest-match-first WordPiece
    song = songs['midi']
    v = MusicVocabulary(pitch_kind='midi', is_wordpiece=True)
    tok = WordPieceMusicTokenizerTrainer(pitch_kind='midi')([song] * 4, vocab_size=len(v) + 150, save=str(tmp_path / 'wp'))
    vocab = tok._tokenizer.get_vocab()
    chars = R.words2chars(R.score2words(song.split(), True, True), v.t2i, R.uni_chars()[:len(v)])
    want = [i for w in chars.split() for i in R.wordpiece_encode_word(w, vocab)]
    assert tok.encode(song) == want


def test_subword_pitch_tables_match_the_reference_expansion(tmp_path, songs):
    """ids2pitches of a sub-word tokenizer = the pitches of the base tokens every id expands to (wordpiece_tokenizer.py:372-379,
    450-452): the (V, 12) pitch-class count table the device metric uses, against the oracle's expansion"""
    from oracle import metrics_ref as MR, subword_ref as R
    from symbolic_music_generation_amd.metrics import pitch_class_hist_table
    from symbolic_music_generation_amd.subword import PairMergeTokenizer, PairMergeTokenizerTrainer
    song = songs['degree']
    tr = PairMergeTokenizerTrainer(pitch_kind='degree')
    want = R.pair_merge_train([song.split()], len(tr.vocab), coverage_ratio=0.9)
    tr([song], coverage_ratio=0.9, save=str(tmp_path / 'pm'))
    tok = PairMergeTokenizer.from_file(str(tmp_path / 'pm'))
    tab = pitch_class_hist_table(tok)
    assert tab.shape == (tok.vocab_size, 12) and tab.dtype == np.uint8
    id2tok = {i: t for t, i in want['added_tok2id'].items()}
    for i in list(range(0, len(tr.vocab), 37)) + sorted(id2tok):
        expanded = id2tok[i].split() if i in id2tok else [tr.vocab.i2t(i)]
        pcs = [p % 12 for p in MR.ids2pitches(expanded)]
        assert tab[i].tolist() == [pcs.count(c) for c in range(12)], (i, expanded)
    ids = tok.encode(song)
    hist = tab[np.asarray(ids)].sum(0)
    pcs = [p % 12 for p in MR.ids2pitches(song.split())]
    assert hist.tolist() == [pcs.count(c) for c in range(12)]
