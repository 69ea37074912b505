"""__graft_entry__.smoke(): one tiny TransfoXL train step + eval forward on cuda:0, checked against the CPU oracle."""
import torch


def run_smoke():
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel   # checker only
    from .transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    kw = dict(vocab_size=1190, n_layer=2, mem_len=64, cutoffs=[], dropout=0.0)
    ref = RefTransfoXLLMHeadModel(RefXLConfig.from_preset('debug', max_length=64, **kw))
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    m = MyTransfoXLLMHeadModel(MyTransfoXLConfig('debug', max_length=64, **kw), device=dev)
    m.load_state_dict(ref.state_dict())
    ids = torch.randint(4, 1190, (2, 64))
    ref.eval(); m.eval()
    with torch.no_grad():
        ro = ref(ids, labels=ids)
    o = m(input_ids=ids.to(dev), labels=ids.to(dev))
    err = (o.prediction_scores.cpu() - ro.prediction_scores).abs().max().item()
    assert err < 3e-2, f'log-prob mismatch vs oracle: {err}'
    assert abs(o.loss.item() - ro.loss.item()) / ro.loss.item() < 1e-2
    m.train(); m.zero_grad()
    o = m(input_ids=ids.to(dev), labels=ids.to(dev))
    m.backward()
    m.engine.optimizer_step(lr=1e-3)
    torch.cuda.synchronize()
    assert torch.isfinite(o.loss).item()
    # the fused attention backward (round 4; dh = 64, mem_len a multiple of 256): one layer's gradients against the oracle's autograd
    kw2 = dict(vocab_size=1190, n_layer=1, mem_len=256, cutoffs=[], dropout=0.0, d_model=128, n_head=2, d_head=64, d_inner=256)
    ref2 = RefTransfoXLLMHeadModel(RefXLConfig.from_preset('debug', max_length=64, **kw2)).train()
    with torch.no_grad():
        for p in ref2.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    m2 = MyTransfoXLLMHeadModel(MyTransfoXLConfig('debug', max_length=64, **kw2), device=dev).train()
    m2.load_state_dict(ref2.state_dict())
    from . import ops
    assert ops.fused_bwd_applies(T=64, dh=64, M=256, Kc=64), 'the smoke shape must take the fused attention backward'
    ro2 = ref2(ids, labels=ids)
    ro2.loss.backward()
    m2.zero_grad()
    o2 = m2(input_ids=ids.to(dev), labels=ids.to(dev))
    m2.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for name in ('transformer.layers.0.dec_attn.qkv_net.weight', 'transformer.layers.0.dec_attn.r_r_bias',
                 'transformer.layers.0.dec_attn.r_w_bias', 'transformer.layers.0.dec_attn.o_net.weight'):
        rg = dict(ref2.named_parameters())[name].grad
        g = m2.engine.g32(name).float().cpu().reshape(rg.shape)
        worst = max(worst, ((g - rg).norm() / (rg.norm() + 1e-12)).item())
    assert worst < 6e-2, f'fused attention backward: gradient mismatch vs oracle: {worst}'
    print(f'smoke ok: max |dlogprob| = {err:.4f}, loss = {o.loss.item():.4f}; fused attention backward worst rel gradient error {worst:.4f}')
