"""The reference's own logged Reformer run shape (notebook/train/reformer.ipynb:1300,1411,2870: base = 12L / 768d, seq 4096, two hash
rounds, batch 17, V = 420) as a stand-alone training loop, for rocprofv3 --kernel-trace --stats (bench.py runs it as the context
leg `published_reformer_base`).  STEPS=5 python scripts/perf_rfbase.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead  # noqa: E402

dev = torch.device('cuda:0')
B = int(os.environ.get('B', '17'))
steps, warm = int(os.environ.get('STEPS', '5')), 2
cfg = MyReformerConfig('base', vocab_size=420, max_position_embeddings=4096, axial_pos_shape=(64, 64))
model = MyReformerModelWithLMHead(cfg, device=dev, seed=77).train()
eng = model.engine
ids = torch.randint(4, 420, (B, 4096), generator=torch.Generator().manual_seed(77)).to(dev)


def step():
    with torch.no_grad():
        eng.zero_grad()
        model(input_ids=ids, labels=ids)
        eng.backward()
        eng.optimizer_step(lr=3e-4, weight_decay=0.01, max_grad_norm=1.0)


for _ in range(warm):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f'Reformer base seq 4096 nh={cfg.num_hashes} batch {B}: {1e3 * dt:.2f} ms per step, {B * 4096 / dt / 1e3:.1f} k tok/s')
