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
    print(f'smoke ok: max |dlogprob| = {err:.4f}, loss = {o.loss.item():.4f}')
