"""Where does the full-size HIP-vs-oracle log-prob difference come from?  Per-layer hidden-state error of the HIP path against
the fp32 oracle at C2 / C3 shapes (B = 1), next to the error of the SAME oracle run with bf16 matmul operands (torch CPU
bf16-storage) -- the precision class the HIP path belongs to -- for weight scales 1x (the reference's init, what bench.py runs)
and 3x (the stress scale of the parity tests).  Prints quantiles; used to set the tolerances in tests/test_fullsize_gpu.py."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
V = 1190


def pair(dev, preset, n_layer, T, M, seed, wscale):
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    torch.manual_seed(seed)
    kw = dict(vocab_size=V, n_layer=n_layer, mem_len=M, max_length=T, cutoffs=[], dropout=0.0)
    ref = RefTransfoXLLMHeadModel(RefXLConfig.from_preset(preset, **kw))
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() > 1 and 'layer_norm' not in n:
                p.mul_(wscale)
            p.copy_(p.to(torch.bfloat16).float())
    m = MyTransfoXLLMHeadModel(MyTransfoXLConfig(preset, **kw), device=dev)
    m.load_state_dict(ref.state_dict())
    return ref.eval(), m.eval()


def bf16_storage_forward(ref, ids):
    """the fp32 oracle with every module output (Linear, LayerNorm, Embedding, positional table, FFN, attention block) rounded to
    bf16 as it is written -- fp32 arithmetic, bf16 storage between operators: the precision class of the HIP path"""
    from torch import nn
    from oracle import transfoxl_ref as R
    kinds = (nn.Linear, nn.LayerNorm, nn.Embedding, R.PositionalEmbedding)
    rnd = lambda mod, inp, out: out.to(torch.bfloat16).float() if torch.is_tensor(out) else out
    hooks = [mod.register_forward_hook(rnd) for mod in ref.modules() if isinstance(mod, kinds)]
    try:
        return ref(ids, labels=ids)
    finally:
        for h in hooks:
            h.remove()


def q(x):
    x = x.flatten().float()
    ks = [0.5, 0.9, 0.99, 0.999, 0.9999]
    idx = [min(int(k * x.numel()), x.numel() - 1) for k in ks]
    s = x.sort().values
    return ' '.join(f'p{k}={s[i].item():.4f}' for k, i in zip(ks, idx)) + f' max={s[-1].item():.4f} mean={x.mean().item():.5f}'


def run(dev, preset, n_layer, T, M, wscale, seed=23):
    ref, m = pair(dev, preset, n_layer, T, M, seed, wscale)
    ids = torch.randint(4, V, (1, T), generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        t = time.time(); ro = ref(ids, labels=ids); t_ref = time.time() - t
        t = time.time()
        ra = bf16_storage_forward(ref, ids)
        t_ac = time.time() - t
        o = m(input_ids=ids.to(dev), labels=ids.to(dev))
    print(f'== {preset} {n_layer}L T={T} weights x{wscale}: oracle {t_ref:.1f}s, bf16-storage oracle {t_ac:.1f}s; loss hip {o.loss.item():.5f} '
          f'ref {ro.loss.item():.5f} bf16-storage {ra.loss.item():.5f}')
    for l in range(n_layer):
        hr = ro.mems[l][:, 0].float()
        hh = o.mems[l][:, 0].float().cpu()
        ha = ra.mems[l][:, 0].float()
        rel = lambda a: ((a - hr).norm() / hr.norm()).item()
        print(f'   layer {l:2d} input: rel err hip {rel(hh):.5f}  bf16-storage {rel(ha):.5f}   |h| rms {hr.pow(2).mean().sqrt().item():.3f}')
    e_h = (o.prediction_scores.float().cpu() - ro.prediction_scores).abs()
    e_a = (ra.prediction_scores.float() - ro.prediction_scores).abs()
    print('   logp err hip     :', q(e_h))
    print('   logp err bf16-storage:', q(e_a))
    top = ro.prediction_scores.topk(8, -1).indices
    print('   logp err on the oracle top-8 tokens: hip', q(e_h.gather(-1, top)), '| bf16-storage', q(e_a.gather(-1, top)))
    agree_h = (o.prediction_scores.cpu().argmax(-1) == ro.prediction_scores.argmax(-1)).float().mean().item()
    agree_a = (ra.prediction_scores.argmax(-1) == ro.prediction_scores.argmax(-1)).float().mean().item()
    print(f'   argmax agreement with the fp32 oracle: hip {agree_h:.4f} bf16-storage {agree_a:.4f}')


if __name__ == '__main__':
    dev = torch.device('cuda:0')
    which = sys.argv[1:] or ['c2', 'c3']
    for w in which:
        for ws in (1.0, 3.0):
            if w == 'c2':
                run(dev, 'small', 6, 1024, 1024, ws)
            elif w == 'c3':
                run(dev, 'base', 12, 2048, 2048, ws)
            elif w == 'c1':
                run(dev, 'debug', 2, 256, 256, ws)
