"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's batch contract:
`tokenizer(toks, padding='max_length', truncation=True)` (musicnlp/preprocess/dataset.py:361; MusicTokenizer pads on the right
with `[PAD]`, music_tokenizer.py:28-40) followed by DataCollatorForLanguageModeling(mlm=False) (musicnlp/trainer/train.py:360):
labels = input_ids.clone(); labels[labels == pad_token_id] = -100.   Only tests/ may import this module."""
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
