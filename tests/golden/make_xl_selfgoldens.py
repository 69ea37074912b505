"""Regression fixture for the TransfoXL restatement itself ("from the build's own TransfoXL restatement: C1-shaped
weights / logits / loss / mems and a 64-token greedy continuation").  NOT an external pin (those are xlnet_*.pt): it freezes what
oracle/transfoxl_ref.py computes today, so that an edit to the oracle or a drift of the HIP path shows up as a diff against
committed numbers rather than against a moving target.

    python tests/golden/make_xl_selfgoldens.py        # writes tests/golden/xl_c1_selfgolden.pt  (weights + inputs + outputs)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel  # noqa: E402

if __name__ == '__main__':
    torch.manual_seed(20260101)
    V = 422                                                    # midi-pitch vocabulary of the reference tokenizer
    cfg = dict(vocab_size=V, n_layer=2, d_model=64, n_head=4, d_head=16, d_inner=256, d_embed=64, mem_len=64, clamp_len=48,
               max_length=64, cutoffs=[200], dropout=0.0, dropatt=0.0)
    rc = RefXLConfig.from_preset('debug', **cfg)
    m = RefTransfoXLLMHeadModel(rc).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1 and 'layer_norm' not in n:
                p.mul_(3.0)
            p.copy_(p.to(torch.bfloat16).float())              # bf16-representable: the HIP path holds the same numbers
    ids = torch.randint(4, V, (2, 128))
    labels = ids.clone(); labels[1, 100:] = -100
    with torch.no_grad():
        o1 = m(ids[:, :64], labels=labels[:, :64])
        o2 = m(ids[:, 64:], mems=o1.mems, labels=labels[:, 64:])
        gen = m.greedy_generate(ids[:, :24], max_length=24 + 64)
    blob = dict(config=cfg, state_dict={k: v.clone() for k, v in m.state_dict().items()}, ids=ids, labels=labels,
                logp1=o1.prediction_scores.half(), logp2=o2.prediction_scores.half(), loss1=o1.loss, loss2=o2.loss,
                mems_last=o2.mems[-1].half(), greedy=gen)
    out = os.path.join(HERE, 'xl_c1_selfgolden.pt')
    torch.save(blob, out)
    print('wrote', out, os.path.getsize(out) // 1024, 'KiB')
