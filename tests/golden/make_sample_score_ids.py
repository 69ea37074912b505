"""Generates tests/golden/sample_score_ids.npz in the BUILD container (where /root/reference exists): the four real token
streams the reference keeps in musicnlp/_sample_score.py (lines 4, 158, 462, 698) are tokenised with this repo's
MusicTokenizer restatement into int16 id arrays.  Only ids (derived data) are stored, never the reference text.
The check that pins the vocabulary restatement: every pitch token of each stream must be in-vocab for its pitch kind;
other OOV tokens must sanitise to their type's rare token.

    python tests/golden/make_sample_score_ids.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from symbolic_music_generation_amd.vocab import MusicTokenizer  # noqa: E402

spec = importlib.util.spec_from_file_location('_sample_score', '/root/reference/musicnlp/_sample_score.py')
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

out = {}
for name, kind in [('sample_full_midi', 'midi'), ('sample_full_step', 'step'), ('sample_full_degree', 'degree'),
                   ('gen_broken', 'degree')]:
    text = getattr(mod, name)
    tok = MusicTokenizer(pitch_kind=kind)
    toks = text.split()
    oov = sorted(set(t for t in toks if t not in tok.vocab))
    # out-of-vocabulary tokens are legal only where the reference sanitises them to the type's rare token
    # (music_vocab.py:883-926); pitches of the stream's own kind must all be in-vocab
    assert all(tok.vocab.type(t) != 'pitch' for t in oov), f'{name}: OOV pitch tokens {oov[:5]}'
    ids = np.array(tok.encode(text), dtype=np.int16)
    assert tok.decode(ids).split() == [tok.vocab.sanitize_rare_token(t) for t in toks]
    print('   sanitised:', oov)
    out[name] = ids
    print(name, kind, len(ids), 'vocab', tok.vocab_size)
np.savez_compressed(os.path.join(os.path.dirname(__file__), 'sample_score_ids.npz'), **out)
