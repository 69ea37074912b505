"""A Transformer-XL training loop at one of the bench's context shapes, stand-alone, for rocprofv3 --kernel-trace --stats:
    SHAPE=c2 python scripts/perf_xl_shape.py        (BASELINE configs[1]: 6L / 512d, T = M = 1024, batch 64)
    SHAPE=pub python scripts/perf_xl_shape.py       (the reference's logged run: base, seq 512, mem 256, V 418, batch 32)"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel  # noqa: E402

dev = torch.device('cuda:0')
shape = os.environ.get('SHAPE', 'c2')
if shape == 'c2':
    size, L, T, M, B, V = 'small', 6, 1024, 1024, 64, 1190
else:
    size, L, T, M, B, V = 'base', 12, 512, 256, 32, 418
B = int(os.environ.get('B', B))
steps, warm = int(os.environ.get('STEPS', '5')), 2
cfg = MyTransfoXLConfig(size, max_length=T, vocab_size=V, n_layer=L, mem_len=M, cutoffs=[])
model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
eng = model.engine
ids = torch.randint(4, V, (B, T), generator=torch.Generator().manual_seed(77)).to(dev)


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
print(f'{shape}: {size} {L}L T={T} M={M} batch {B}: {1e3 * dt:.2f} ms per step, {B * T / dt / 1e3:.1f} k tok/s')
