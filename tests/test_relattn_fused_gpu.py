"""mxl_relattn_bwd_fused (round 4: the attention backward as one pass over the score cells, no dG tensor) against the fp32
autograd of the dense statement (oracle/relattn_ref.py, the form pinned on HF XLNet's rel_attn_core) and against the
three-kernel path it replaces.  Tolerance as tests/test_ops_gpu.py::test_relattn_bwd: 2e-2 relative (Frobenius) per tensor --
bf16 P / dS operands and bf16 outputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def _inputs(B, T, H, dh, M, Kc, seed, scale=0.8):
    torch.manual_seed(seed)
    d = H * dh
    qkv = bf(torch.randn(B, Kc, 3 * d) * scale)
    rd = bf(torch.randn(M, d) * scale)
    rwb, rrb = torch.randn(H, dh) * 0.5, torch.randn(H, dh) * 0.5
    dout = bf(torch.randn(B, T, d))
    return qkv, rd, rwb, rrb, dout


def _run_fused(dev, qkv, rd, rwb, rrb, dout, B, T, H, dh, M, Kc, records_from_forward=False):
    from symbolic_music_generation_amd import ops
    d = H * dh
    qkv_d, rd_d, do_d = qkv.to(dev), rd.to(dev), dout.to(dev)
    rwb_d, rrb_d = rwb.to(dev), rrb.to(dev)
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=dev)
    st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d,
              o_bs=T * d, o_rs=d)
    qv, kv, vv = qkv_d[:, Kc - T:, :d], qkv_d[:, :, d:2 * d], qkv_d[:, :, 2 * d:]
    zero_mem = Kc < M + T
    oph = torch.full((B, T, d), float('nan'), device=dev, dtype=torch.bfloat16) if zero_mem else None
    mph = torch.full((B, H, T), float('nan'), device=dev) if zero_mem else None
    # the phantom cells' dRd kernel reads per-tile records: written by the forward (the training path) or by its own prep pass
    ph = (torch.full((int(ops.lib().mxl_relattn_drd_phantom_ws_bytes(B, T, H)),), 0xFF, device=dev, dtype=torch.uint8)
          if (records_from_forward and zero_mem) else None)
    ops.relattn_fwd(qv, kv, vv, rd_d, rwb_d, rrb_d, out, lse, oph=oph, mph=mph, oph_all=True, ph_buf=ph, **st)
    dqkv = torch.full((B, Kc, 3 * d), float('nan'), device=dev, dtype=torch.bfloat16)
    delta = torch.zeros(B, H, T, device=dev)
    d_rwb, d_rrb = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
    d_rd = torch.zeros(M, d, device=dev)
    qr_buf = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
    ws = torch.full((ops.relattn_bwd_fused_ws_numel(B, T, H, dh, M),), float('nan'), device=dev)
    ops.relattn_bwd_fused(qv, kv, vv, rd_d, rwb_d, rrb_d, out, do_d, lse, delta, dqkv[:, Kc - T:, :d], dqkv[:, :, d:2 * d],
                          dqkv[:, :, 2 * d:], d_rd, d_rwb, d_rrb, ws, qr_buf, dq_bs=Kc * 3 * d, dq_rs=3 * d, dkv_bs=Kc * 3 * d,
                          dkv_rs=3 * d, oph=oph, mph=mph, ph_buf=ph, ph_ready=ph is not None, **st)
    torch.cuda.synchronize()
    return dict(out=out, lse=lse, dq=dqkv[:, Kc - T:, :d], dk=dqkv[:, :, d:2 * d], dv=dqkv[:, :, 2 * d:], d_rd=d_rd, d_rwb=d_rwb,
                d_rrb=d_rrb)


FUSED_CASES = [
    # B, T, H, dh, M, Kc, name          (dh = 64, T % 32 == 0, M % 32 == 0, (T - Kc) % 64 == 0)
    (2, 512, 2, 64, 512, 512, 'zero mems, T = M (the training shape in small)'),
    (1, 256, 2, 64, 256, 512, 'full carried memory (Kc = M + T)'),
    (3, 768, 1, 64, 1024, 768 + 192, 'partial memory, T < M'),
    (1, 1024, 2, 64, 768, 1024 + 64, 'partial memory, T > M'),
    (2, 320, 3, 64, 256, 320, 'T not a multiple of the key block'),
    (1, 64, 1, 64, 256, 64, 'T < M, zero mems: mostly phantom distances'),
    (1, 1280, 1, 64, 512, 1280, 'T = 2.5 M, zero mems: queries past the memory window'),
    (1, 96, 2, 64, 512, 96 + 512, 'short segment over a long carried memory'),
    # round 6: M need only be a multiple of 32 (the reference's `small` preset has mem_len 128: musicnlp/models/transformer_xl.py:16-34)
    (2, 1024, 2, 64, 128, 1024, 'mem_len 128 (the reference small preset), zero mems'),
    (1, 512, 2, 64, 128, 512 + 128, 'mem_len 128, full carried memory'),
    (2, 512, 1, 64, 384, 512, 'mem_len 384: a window of one and a half key blocks'),
    (1, 256, 2, 64, 96, 256 + 64, 'mem_len 96, partial memory'),
    (1, 640, 1, 64, 608, 640, 'mem_len 608 = 19 distance blocks, zero mems'),
]


@pytest.mark.parametrize('B,T,H,dh,M,Kc,name', FUSED_CASES)
def test_relattn_bwd_fused_vs_autograd(dev, B, T, H, dh, M, Kc, name):
    from oracle.relattn_ref import relattn_dense
    d = H * dh
    qkv, rd, rwb, rrb, dout = _inputs(B, T, H, dh, M, Kc, seed=T * 3 + M + Kc)
    q = qkv[:, Kc - T:, :d].float().view(B, T, H, dh).clone().requires_grad_(True)
    k = qkv[:, :, d:2 * d].float().view(B, Kc, H, dh).clone().requires_grad_(True)
    v = qkv[:, :, 2 * d:].float().view(B, Kc, H, dh).clone().requires_grad_(True)
    rdr = rd.float().view(M, H, dh).clone().requires_grad_(True)
    rwbr, rrbr = rwb.clone().requires_grad_(True), rrb.clone().requires_grad_(True)
    ref_out, _ = relattn_dense(q, k, v, rdr, rwbr, rrbr, M)
    ref_out.backward(dout.float().view(B, T, H, dh))
    g = _run_fused(dev, qkv, rd, rwb, rrb, dout, B, T, H, dh, M, Kc)
    errs = {}
    for nm, got, ref in [('dq', g['dq'].float().cpu().view(B, T, H, dh), q.grad),
                         ('dk', g['dk'].float().cpu().view(B, Kc, H, dh), k.grad),
                         ('dv', g['dv'].float().cpu().view(B, Kc, H, dh), v.grad),
                         ('d_rd', g['d_rd'].cpu().view(M, H, dh), rdr.grad), ('d_rwb', g['d_rwb'].cpu(), rwbr.grad),
                         ('d_rrb', g['d_rrb'].cpu(), rrbr.grad)]:
        assert torch.isfinite(got).all(), f'{name}: {nm} holds non-finite values (unwritten output?)'
        errs[nm] = rel_err(got, ref)
    assert all(e < 2e-2 for e in errs.values()), f'{name}: {errs}'


def test_relattn_bwd_fused_matches_three_kernel_path_at_c3_shape(dev):
    """C3 layer shape (H = 12, T = M = 2048, zero mems), B = 2: the fused pass and the query-owner / key-owner / dRd kernels it
    replaces compute the same gradients from the same forward (both round P and dS to bf16 once; the sums differ in order)."""
    from symbolic_music_generation_amd import ops
    B, T, H, dh, M, Kc = 2, 2048, 12, 64, 2048, 2048
    d = H * dh
    qkv, rd, rwb, rrb, dout = _inputs(B, T, H, dh, M, Kc, seed=11, scale=0.5)
    g = _run_fused(dev, qkv, rd, rwb, rrb, dout, B, T, H, dh, M, Kc)
    qkv_d, rd_d, do_d = qkv.to(dev), rd.to(dev), dout.to(dev)
    rwb_d, rrb_d = rwb.to(dev), rrb.to(dev)
    st = dict(B=B, T=T, H=H, dh=dh, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d,
              o_bs=T * d, o_rs=d)
    qv, kv, vv = qkv_d[:, Kc - T:, :d], qkv_d[:, :, d:2 * d], qkv_d[:, :, 2 * d:]
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device=dev)
    ops.relattn_fwd(qv, kv, vv, rd_d, rwb_d, rrb_d, out, lse, **st)
    assert torch.equal(out, g['out']) and torch.equal(lse, g['lse'])      # the value-sum does not touch the forward's outputs
    dqkv = torch.zeros(B, Kc, 3 * d, device=dev, dtype=torch.bfloat16)
    delta = torch.zeros(B, H, T, device=dev)
    dg = torch.empty(B, H, T, M, device=dev, dtype=torch.bfloat16)
    d_rwb, d_rrb = torch.zeros(H, dh, device=dev), torch.zeros(H, dh, device=dev)
    d_rd = torch.zeros(M, d, device=dev)
    qr_buf = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
    ops.relattn_bwd(qv, kv, vv, rd_d, rwb_d, rrb_d, out, do_d, lse, delta, dqkv[:, Kc - T:, :d], dqkv[:, :, d:2 * d],
                    dqkv[:, :, 2 * d:], dg, d_rwb, d_rrb, dq_bs=Kc * 3 * d, dq_rs=3 * d, dkv_bs=Kc * 3 * d, dkv_rs=3 * d,
                    d_rd=d_rd, qr_buf=qr_buf, **st)
    torch.cuda.synchronize()
    errs = {'dq': rel_err(g['dq'], dqkv[:, Kc - T:, :d]), 'dk': rel_err(g['dk'], dqkv[:, :, d:2 * d]),
            'dv': rel_err(g['dv'], dqkv[:, :, 2 * d:]), 'd_rd': rel_err(g['d_rd'], d_rd), 'd_rwb': rel_err(g['d_rwb'], d_rwb),
            'd_rrb': rel_err(g['d_rrb'], d_rrb)}
    assert all(e < 1e-2 for e in errs.values()), errs


@pytest.mark.parametrize('B,T,H,dh,M,Kc', [(2, 512, 2, 64, 512, 512), (1, 1280, 1, 64, 512, 1280), (3, 768, 1, 64, 1024, 768 + 192)])
def test_phantom_records_from_the_forward_match_the_prep_pass(dev, B, T, H, dh, M, Kc):
    """mxl_relattn_fwd_phantom2(..., ph_ws) writes the records mxl_relattn_drd_phantom reads (scaled q + r_r_bias rows in LDS image
    order, -lse2); mxl_relattn_drd_phantom_prep builds the same records from q and lse.  The rows are the same bf16 values; -lse2
    differs by one rounding of lse, so d_rd agrees to float-atomic order + 1e-5."""
    x = _inputs(B, T, H, dh, M, Kc, seed=5)
    a = _run_fused(dev, *x, B, T, H, dh, M, Kc, records_from_forward=False)
    b = _run_fused(dev, *x, B, T, H, dh, M, Kc, records_from_forward=True)
    for k in ('dq', 'dk', 'dv', 'd_rwb', 'd_rrb'):
        assert torch.allclose(a[k].float(), b[k].float(), rtol=1e-4, atol=1e-5), k
    ref = a['d_rd'].float()
    assert (ref - b['d_rd'].float()).norm() <= 2e-4 * ref.norm()


def test_relattn_bwd_fused_is_reproducible_in_dq_dk_dv(dev):
    """dq (slab sums in a fixed order), dk and dv (owned by one workgroup) are bit-identical from run to run; d_rd and the bias
    gradients are float-atomic sums and agree to summation order."""
    B, T, H, dh, M, Kc = 2, 512, 2, 64, 512, 512
    qkv, rd, rwb, rrb, dout = _inputs(B, T, H, dh, M, Kc, seed=5)
    a = _run_fused(dev, qkv, rd, rwb, rrb, dout, B, T, H, dh, M, Kc)
    b = _run_fused(dev, qkv, rd, rwb, rrb, dout, B, T, H, dh, M, Kc)
    for nm in ('dq', 'dk', 'dv'):
        assert torch.equal(a[nm], b[nm]), nm
    for nm in ('d_rd', 'd_rwb', 'd_rrb'):
        assert rel_err(a[nm], b[nm]) < 1e-5, nm


def test_relattn_bwd_fused_rejects_shapes_it_does_not_take(dev):
    from symbolic_music_generation_amd import ops
    from symbolic_music_generation_amd._lib import lib
    assert not ops.fused_bwd_applies(T=200, dh=64, M=256, Kc=200)
    assert not ops.fused_bwd_applies(T=256, dh=32, M=256, Kc=256)
    assert ops.fused_bwd_applies(T=256, dh=64, M=320, Kc=256)               # (round 6: any multiple of 32 distances)
    assert not ops.fused_bwd_applies(T=256, dh=64, M=330, Kc=256)
    assert not ops.fused_bwd_applies(T=256, dh=64, M=256, Kc=256 + 32)      # first stored key not on a 64-key tile boundary
    assert ops.fused_bwd_applies(T=2048, dh=64, M=2048, Kc=2048)
    assert not ops.fused_bwd_applies(T=2048, dh=64, M=16384, Kc=2048)       # the phantom-cell kernel tables 32 distance blocks
    assert not ops.fused_bwd_applies(T=2048, dh=64, M=2048, Kc=2048, B=8192, H=64)      # 32-bit offsets of its buffer addressing
    assert lib().mxl_relattn_bwd_fused_ws_bytes(2, 256, 2, 32, 256) == 0


@pytest.mark.parametrize('with_mem', [False, True])
def test_reference_small_preset_default_memory_trains_on_the_fused_pass(dev, monkeypatch, with_mem):
    """the reference's `small` preset with its DEFAULT mem_len = max(128, 1024 // 8) = 128 (musicnlp/models/transformer_xl.py:16-34;
    d = 512, 8 heads of 64), two layers: the engine picks the fused backward (until round 6: M % 256 != 0 -> the three-kernel path);
    its gradients are those of the three-kernel path from the same forward (1e-2 per tensor) and match the oracle's autograd as
    closely as that path does (at this test's weight scale -- 3 x the initialisation -- both sit at 6-10 % on every tensor)"""
    from tests.test_xl_model_gpu import _pair
    from symbolic_music_generation_amd import ops
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
    assert MyTransfoXLConfig('small', vocab_size=1190).mem_len == 128          # the preset's own default
    B, T, M, d = 2, 384, 128, 512
    ids = torch.randint(4, 1190, (B, T), generator=torch.Generator().manual_seed(1))
    lab = ids.clone(); lab[0, 300:] = -100
    mems_r = [torch.randn(M, B, d, generator=torch.Generator().manual_seed(2 + i)).to(torch.bfloat16).float() for i in range(2)] if with_mem else None

    def run(legacy):
        monkeypatch.setenv('MXL_NO_FUSED_BWD', '1' if legacy else '0')
        ref, m = _pair(dev, preset='small', n_layer=2, mem_len=128, max_length=384, seed=7)
        assert m.config.mem_len == 128 and m.config.d_head == 64
        m.train()
        m.zero_grad()
        o = m(input_ids=ids.to(dev), labels=lab.to(dev), mems=[x.to(dev) for x in mems_r] if with_mem else None)
        m.backward()
        torch.cuda.synchronize()
        assert m.engine._last.fused_bwd == (not legacy)
        return ref, o.loss.item(), {n: m.engine.g32(n).float().cpu().clone() for n, _ in ref.named_parameters() if n != 'crit.out_layers.0.weight'}

    assert ops.fused_bwd_applies(T=T, dh=64, M=M, Kc=T + (M if with_mem else 0), B=B, H=8)
    ref, loss_f, gf = run(False)
    _, loss_l, gl = run(True)
    assert abs(loss_f - loss_l) <= 1e-6 * abs(loss_l)      # (the same forward kernels; the loss reduction sums with float atomics)
    ref.train()
    ro = ref(ids, labels=lab, mems=mems_r)
    ro.loss.backward()
    assert abs(loss_f - ro.loss.item()) / ro.loss.item() < 1e-2
    bad = {}
    for name, p in ref.named_parameters():
        if name not in gf:
            continue
        e_paths = ((gf[name] - gl[name]).norm() / (gl[name].norm() + 1e-12)).item()
        e_f = ((gf[name] - p.grad).norm() / (p.grad.norm() + 1e-12)).item()
        e_l = ((gl[name] - p.grad).norm() / (p.grad.norm() + 1e-12)).item()
        if e_paths > 1e-2 or e_f > max(0.12, 1.1 * e_l):
            bad[name] = (e_paths, e_f, e_l)
    assert not bad, f'(fused vs three-kernel, fused vs oracle, three-kernel vs oracle): {bad}'
