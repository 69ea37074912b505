"""Device evaluation metrics vs the CPU restatement (oracle/metrics_ref.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import metrics_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def test_argmax_rows_first_maximum(dev):
    from symbolic_music_generation_amd.metrics import max_out_logits
    torch.manual_seed(0)
    x = torch.randn(3, 37, 1190, device=dev)
    x[0, 0, 5] = x[0, 0, 900] = 50.0          # tie: the first index wins
    x[1, 2, :] = -float('inf'); x[1, 2, 77] = -3.0
    ids = max_out_logits(x)
    assert ids.shape == (3, 37) and ids.dtype == torch.int64
    assert torch.equal(ids.cpu(), torch.from_numpy(x.cpu().numpy().argmax(-1)))
    assert ids[0, 0].item() == 5 and ids[1, 2].item() == 77


@pytest.mark.parametrize('kind,shifted', [('midi', False), ('degree', False), ('degree', True)])
def test_compute_metrics_matches_oracle(dev, kind, shifted):
    from symbolic_music_generation_amd.metrics import ComputeMetrics
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    tok = MusicTokenizer(pitch_kind=kind)
    V = len(tok.vocab)
    rng = np.random.default_rng(5)
    B, T = 6, 333
    labels = rng.integers(0, V, size=(B, T))
    labels[rng.random((B, T)) < 0.15] = -100
    labels[3, 200:] = -100                                     # a padded tail
    preds = rng.integers(0, V, size=(B, T - 1 if shifted else T))
    agree = rng.random(preds.shape) < 0.4                      # make a good share of the predictions correct
    nxt = labels[:, 1:]
    tgt = preds[:, :T - 1]
    tgt[agree[:, :T - 1] & (nxt >= 0)] = nxt[agree[:, :T - 1] & (nxt >= 0)]
    key_scores = np.zeros((B, 24))
    for b in range(B):
        for o in rng.choice(24, size=3, replace=False):
            key_scores[b, o] = rng.random() + 0.1
    cm = ComputeMetrics(tok, mode='vanilla', clm_pred_shifted=shifted)
    got = cm((torch.from_numpy(preds).to(dev), torch.from_numpy(labels).to(dev), key_scores))
    ref_ikr = R.ikr(preds, labels, tok.vocab.i2t, key_scores=key_scores, mode='vanilla', clm_pred_shifted=shifted)
    ref_acc = R.ntp_acc(preds, labels, clm_pred_shifted=shifted)
    assert got['ikr'] == pytest.approx(ref_ikr, abs=1e-12)
    assert got['ntp_acc'] == pytest.approx(ref_acc, abs=1e-12)
    # 'ins-key': key token at the third label position
    labels2 = labels.copy()
    for b in range(B):
        labels2[b, 2] = tok.vocab.t2i('Key_' + R.KEY_STRS[(5 * b + 1) % 24])
    cm2 = ComputeMetrics(tok, mode='ins-key', clm_pred_shifted=shifted)
    got2 = cm2((torch.from_numpy(preds).to(dev), torch.from_numpy(labels2).to(dev)))
    assert got2['ikr'] == pytest.approx(R.ikr(preds, labels2, tok.vocab.i2t, mode='ins-key', clm_pred_shifted=shifted), abs=1e-12)


def test_metrics_on_reference_sample_scores(dev):
    """the reference's real token streams (tests/golden/sample_score_ids.npz): teacher-forced 'perfect' predictions give
    accuracy 1 and the in-key ratio of the piece itself, identical on device and in the restatement"""
    from symbolic_music_generation_amd.metrics import ComputeMetrics
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'sample_score_ids.npz'))
    tok = MusicTokenizer(pitch_kind='degree')
    name = [k for k in z.files if 'degree' in k][0]
    ids = z[name].astype(np.int64)[:2048][None, :]
    labels = ids.copy()
    preds = np.concatenate([ids[:, 1:], ids[:, -1:]], axis=1)       # pred[j] = label[j+1]
    ks = np.zeros((1, 24)); ks[0, R.KEY_STRS.index('CMajor')] = 1.0; ks[0, R.KEY_STRS.index('AMinor')] = 0.5
    cm = ComputeMetrics(tok, mode='vanilla')
    got = cm((torch.from_numpy(preds).to(dev), torch.from_numpy(labels).to(dev), ks))
    assert got['ntp_acc'] == 1.0
    assert got['ikr'] == pytest.approx(R.ikr(preds, labels, tok.vocab.i2t, key_scores=ks, mode='vanilla'), abs=1e-12)
    assert 0.0 < got['ikr'] <= 1.0
