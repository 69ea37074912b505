"""diagnose the two bias gradients that miss the oracle at the C3 layer shapes (2-layer model, T = M = 2048)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_fullsize_gpu import _oracle_pair, T, M, V
dev = torch.device('cuda:0')
L = int(os.environ.get('L', 2)); Tn = int(os.environ.get('TN', T)); preset = os.environ.get('PRESET', 'base')
ref, m = _oracle_pair(dev, preset, L, Tn, Tn, seed=41, wscale=1.0)
ref.train(); m.train()
g = torch.Generator().manual_seed(42)
ids = torch.randint(4, V, (1, Tn), generator=g)
lab = ids.clone(); lab[0, Tn - 100:] = -100
ro = ref(ids, labels=lab); ro.loss.backward()
m.zero_grad(); o = m(input_ids=ids.to(dev), labels=lab.to(dev)); m.backward(); torch.cuda.synchronize()
print('loss', o.loss.item(), ro.loss.item())
rows = []
for n, p in ref.named_parameters():
    if n == 'crit.out_layers.0.weight': continue
    a = m.engine.g32(n).float().cpu().reshape(p.grad.shape); b = p.grad
    e = ((a - b).norm() / (b.norm() + 1e-12)).item()
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    rows.append((e, cos, n, b.norm().item(), a.norm().item()))
for r in sorted(rows, reverse=True)[:12]:
    print(f'{r[2]:55s} rel {r[0]:.4f} cos {r[1]:.5f} |ref| {r[3]:.4e} |hip| {r[4]:.4e}')
for n in ['crit.out_layers.0.bias', f'transformer.layers.{L-1}.pos_ff.layer_norm.bias']:
    a = m.engine.g32(n).float().cpu().flatten(); b = dict(ref.named_parameters())[n].grad.flatten()
    d = a - b
    print(n, 'diff mean', d.mean().item(), 'diff std', d.std().item(), 'ref mean', b.mean().item(), 'ref std', b.std().item(),
          'hip mean', a.mean().item(), 'slope', (a @ b / (b @ b)).item())
    print('   first 8 ref', b[:8].tolist()); print('   first 8 hip', a[:8].tolist())
