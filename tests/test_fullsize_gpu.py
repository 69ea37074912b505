"""Size-independent properties at the FULL benchmark sizes (BASELINE.json configs[2] = SURVEY C3: 12L / 768d / H12 / dh64, T = M =
2048, V = 1190; token matrices of 32768 rows), where the CPU oracle is out of reach: checksums for the GEMMs, normalisation
and causality for the attention kernel, normalisation / causality / segmentation invariance / batch equivariance for the
whole model.  The same for the Reformer at configs[3] = SURVEY C4 (T = 8192): sortedness / permutation of the LSH sort, softmax mass
and the exact reach of a key in the chunked attention, and whole-model equivariance / causality."""
import pytest
import torch

pytestmark = pytest.mark.gpu

T = M = 2048
D, H, DH, F, V = 768, 12, 64, 3072, 1190
NTOK = 16 * T


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize('name,O,K', [('qkv', 3 * D, D), ('o', D, D), ('ffn1', F, D), ('ffn2', D, F)])
def test_gemm_checksums_at_c3_shapes(dev, name, O, K):
    """sum_m C[m, :] == (sum_m A[m, :]) W^T for y = x W^T, dX = dY W and dW = dY^T X (row / column checksums computed in
    fp64 from the same bf16 operands): a wrong tile, a dropped K-step or a lost atomic shows up at any size"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(K + O)
    x = (torch.randn(NTOK, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(O, K, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(NTOK, O, device=dev, dtype=torch.float32)
    ops.gemm(x, w, y, NTOK, O, K, flags=ops.GEMM_OUT_F32)                       # old kernel (fp32 out)
    yb = torch.empty(NTOK, O, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, w, yb, NTOK, O, K)                                              # persistent DMA kernel (bf16 out)
    col = x.double().sum(0) @ w.double().t()                                    # column checksum of Y
    assert _rel(y.double().sum(0), col) < 1e-5
    assert _rel(yb.double().sum(0), col) < 2e-3                                 # bf16 rounding of 32768 outputs per column
    row = x.double() @ w.double().sum(0)                                        # row checksum of Y
    assert _rel(yb.double().sum(1), row) < 2e-3
    # dX = dY W (through the [in][out] copy, as the engine does) and dW = dY^T X (split-K atomics)
    wt = torch.empty(K, O, device=dev, dtype=torch.bfloat16)
    ops.transpose(w, wt, O, K)
    dx = torch.empty(NTOK, K, device=dev, dtype=torch.bfloat16)
    ops.gemm(yb, wt, dx, NTOK, K, O)
    assert _rel(dx.double().sum(0), yb.double().sum(0) @ w.double()) < 2e-3
    dw = torch.zeros(O, K, device=dev, dtype=torch.float32)
    ops.gemm(yb, x, dw, O, K, NTOK, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=4)
    assert _rel(dw.double().sum(0), yb.double().sum(1) @ x.double()) < 1e-5     # column checksum: (sum_o dY[:, o])^T X
    assert _rel(dw.double().sum(1), yb.double().t() @ x.double().sum(1)) < 1e-5


def _attn_inputs(dev, B, Kc, seed=0):
    torch.manual_seed(seed)
    d = H * DH
    qkv = torch.randn(B, Kc, 3 * d, device=dev).to(torch.bfloat16)
    rd = torch.randn(M, d, device=dev).to(torch.bfloat16)
    rwb = torch.randn(H, DH, device=dev) * 0.1
    rrb = torch.randn(H, DH, device=dev) * 0.1
    st = dict(B=B, T=T, H=H, dh=DH, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d, o_bs=T * d, o_rs=d)
    return qkv, rd, rwb, rrb, st, d


def test_attention_normalisation_and_causality_at_c3(dev):
    from symbolic_music_generation_amd import ops
    B, Kc = 2, T + M
    qkv, rd, rwb, rrb, st, d = _attn_inputs(dev, B, Kc)
    qkv[:, :, 2 * d:] = 1.0                                   # V = 1: the output is the softmax mass, exactly 1 per (query, head)
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    run = lambda: ops.relattn_fwd(qkv[:, Kc - T:, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:], rd, rwb, rrb, out, lse, **st)
    run()
    assert (out.float() - 1.0).abs().max().item() < 8e-3      # bf16 P, bf16 output
    # causality / window: perturbing key position p (row p + M of the Kc stored rows) changes queries i in [p, p + M - 1] only
    qkv2, *_ = _attn_inputs(dev, B, Kc, seed=0)
    base = torch.zeros_like(out)
    ops.relattn_fwd(qkv2[:, Kc - T:, :d], qkv2[:, :, d:2 * d], qkv2[:, :, 2 * d:], rd, rwb, rrb, base, lse, **st)
    p = 700
    qkv2[:, p + M, d:] += 1.0                                 # k and v of key position p
    pert = torch.zeros_like(out)
    ops.relattn_fwd(qkv2[:, Kc - T:, :d], qkv2[:, :, d:2 * d], qkv2[:, :, 2 * d:], rd, rwb, rrb, pert, lse, **st)
    same = (pert == base).view(B, T, -1).all(-1).all(0)       # per query position, over batch / heads / channels
    assert same[:p].all() and (~same[p:p + 64]).any()         # earlier queries bit-identical, the next ones see the change
    # a key in the memory region leaves the window of late queries: position q0 = -M + 100 is visible to i <= q0 + M - 1 = 99
    qkv3, *_ = _attn_inputs(dev, B, Kc, seed=0)
    qkv3[:, 100, d:] += 1.0
    pert2 = torch.zeros_like(out)
    ops.relattn_fwd(qkv3[:, Kc - T:, :d], qkv3[:, :, d:2 * d], qkv3[:, :, 2 * d:], rd, rwb, rrb, pert2, lse, **st)
    same2 = (pert2 == base).view(B, T, -1).all(-1).all(0)
    assert same2[100:].all() and (~same2[:100]).any()


@pytest.mark.parametrize('B', [32, 64])
def test_attention_backward_batch_replication_at_bench_batch(dev, B):
    """bench.py's per-GPU batch (64 sequences: dG holds 3.2 G elements -- past 2^31, the 64-bit offset paths -- and the fused qkv
    gradient 604 M; 32 was round 1's default): with the same
    sequence in every batch slot the per-sequence outputs (dq, dk, dv, dG, delta) of the last slot equal those of the first
    bit for bit (owner-computes kernels, no cross-sequence accumulation), and the batch-summed ones (d_rd, bias gradients)
    are B x a B = 1 call up to fp32 atomic ordering -- an index that wrapped at these offsets would break either."""
    from symbolic_music_generation_amd import ops
    Kc = T
    d = H * DH

    def run(B):
        qkv1, rd, rwb, rrb, _, _ = _attn_inputs(dev, 1, Kc, seed=3)
        qkv = (qkv1 * 0.5).expand(B, Kc, 3 * d).contiguous()
        torch.manual_seed(5)
        dout = torch.randn(1, T, d, device=dev).to(torch.bfloat16).expand(B, T, d).contiguous()
        st = dict(B=B, T=T, H=H, dh=DH, M=M, Kc=Kc, q_bs=Kc * 3 * d, q_rs=3 * d, kv_bs=Kc * 3 * d, kv_rs=3 * d, rd_rs=d,
                  o_bs=T * d, o_rs=d)
        out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
        qv, kv, vv = qkv[:, Kc - T:, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        ops.relattn_fwd(qv, kv, vv, rd, rwb, rrb, out, lse, **st)
        dqkv = torch.zeros(B, Kc, 3 * d, device=dev, dtype=torch.bfloat16)
        delta = torch.zeros(B, H, T, device=dev)
        dg = torch.full((B, H, T, M), float('nan'), device=dev, dtype=torch.bfloat16)
        d_rwb, d_rrb = torch.zeros(H, DH, device=dev), torch.zeros(H, DH, device=dev)
        d_rd = torch.zeros(M, d, device=dev)
        qr_buf = torch.empty(B, T, d, device=dev, dtype=torch.bfloat16)
        ops.relattn_bwd(qv, kv, vv, rd, rwb, rrb, out, dout, lse, delta, dqkv[:, Kc - T:, :d], dqkv[:, :, d:2 * d],
                        dqkv[:, :, 2 * d:], dg, d_rwb, d_rrb, dq_bs=Kc * 3 * d, dq_rs=3 * d, dkv_bs=Kc * 3 * d, dkv_rs=3 * d,
                        d_rd=d_rd, qr_buf=qr_buf, **st)
        torch.cuda.synchronize()
        return out, lse, dqkv, delta, dg, d_rd, d_rwb, d_rrb

    out, lse, dqkv, delta, dg, d_rd, d_rwb, d_rrb = run(B)
    # mode R at M % 256 == 0: the (32 queries x 256 distances) blocks of dG that lie on phantom distances only are not stored
    # (the dRd contraction rebuilds them); everything else is written
    ii = torch.arange(T, device=dev)[:, None] | 31
    dd = torch.arange(M, device=dev)[None, :] & ~255
    unwritten = (dd > ii).expand(H, T, M)
    for b in (0, B - 1):
        assert torch.equal(torch.isnan(dg[b].float()), unwritten), 'wrong set of dG blocks left to the recompute'
    dg = dg.view(torch.int16)                    # bit patterns: the NaN filler compares equal to itself
    for nm, t in [('out', out), ('lse', lse), ('dqkv', dqkv), ('delta', delta), ('dg', dg)]:
        for b in (1, B // 2, B - 1):
            assert torch.equal(t[b], t[0]), f'{nm}: batch slot {b} differs from slot 0'
    assert dqkv[0].float().abs().sum().item() > 0
    out1, lse1, dqkv1, delta1, dg1, d_rd1, d_rwb1, d_rrb1 = run(1)
    assert torch.equal(dqkv[B - 1], dqkv1[0]) and torch.equal(dg[B - 1], dg1.view(torch.int16)[0]) and torch.equal(out[B - 1], out1[0])
    for nm, a, b in [('d_rd', d_rd, d_rd1), ('d_rwb', d_rwb, d_rwb1), ('d_rrb', d_rrb, d_rrb1)]:
        assert _rel(a, B * b) < 1e-4, f'{nm}: {_rel(a, B * b)}'


def test_model_properties_at_c3(dev):
    """the 12L / 768d model at T = M = 2048: log-prob normalisation, causality, segmentation invariance, batch equivariance"""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    cfg = MyTransfoXLConfig('base', max_length=T, vocab_size=V, n_layer=12, mem_len=M, cutoffs=[])
    m = MyTransfoXLLMHeadModel(cfg, device=dev, seed=5).eval()
    with torch.no_grad():                                     # 3x the init scale: non-trivial attention / softmax
        sd = {k: (v * 3.0 if v.dim() > 1 and 'layer_norm' not in k else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    torch.manual_seed(1)
    ids = torch.randint(4, V, (2, T), device=dev)
    o = m(input_ids=ids)
    lp = o.prediction_scores.float()
    assert lp.shape == (2, T, V)
    assert (lp.exp().sum(-1) - 1.0).abs().max().item() < 2e-3
    # causality: changing token t leaves every earlier position bit-identical
    t = 1500
    ids2 = ids.clone(); ids2[:, t] = (ids2[:, t] + 7) % (V - 4) + 4
    lp2 = m(input_ids=ids2).prediction_scores.float()
    assert torch.equal(lp2[:, :t], lp[:, :t]) and not torch.equal(lp2[:, t:], lp[:, t:])
    # batch equivariance (bit-exact: no cross-sequence arithmetic)
    lp_sw = m(input_ids=ids.flip(0)).prediction_scores.float()
    assert torch.equal(lp_sw.flip(0), lp)
    # segmentation invariance: two 1024-token segments with carried mems == one 2048-token pass (bf16 noise only)
    o1 = m(input_ids=ids[:, :1024])
    o2 = m(input_ids=ids[:, 1024:], mems=o1.mems)
    seg = torch.cat([o1.prediction_scores, o2.prediction_scores], 1).float()
    assert (seg - lp).abs().max().item() < 8e-2
    assert (seg - lp).abs().mean().item() < 1e-2        # log-probs are O(10): 1e-3 relative, 12 layers of bf16 activations


# ----------------------------------------------------------------------------------------------------------------------
# Reformer at SURVEY C4 sizes (BASELINE.json configs[3]): 6L (3 local + 3 LSH) / 512d / H8 / dh64, T = 8192, buckets [16, 16]
# ----------------------------------------------------------------------------------------------------------------------
RT, RD, RH = 8192, 512, 8


def test_lsh_sort_is_the_stable_permutation_at_c4(dev):
    """64 (sequence, head) rows of 8192 keys in 256 buckets: the result is a permutation, sorted by bucket, ties in index order
    (= argsort(S * bucket + index), HF's sort key), and sorted_pos is the position of each slot"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(0)
    BH, S, NB = 64, RT, 256
    bk = torch.randint(0, NB, (BH, S), device=dev, dtype=torch.int32)
    bk[3] = 7                                                      # one bucket holds everything
    bk[4] = torch.arange(S, device=dev, dtype=torch.int32) % NB    # perfectly interleaved
    sidx = torch.empty(BH, S, device=dev, dtype=torch.int32)
    spos = torch.empty_like(sidx)
    ops.lsh_sort(bk, sidx, spos, BH, S, RT, NB)
    si = sidx.long()
    assert torch.equal(si.sort(-1).values, torch.arange(S, device=dev).expand(BH, S))        # a permutation
    sb = bk.long().gather(1, si)
    key = sb * S + si
    assert (key[:, 1:] > key[:, :-1]).all()                        # bucket ascending, index ascending inside a bucket
    assert torch.equal(spos.long(), si % RT)
    assert torch.equal(si, torch.argsort(bk.long() * S + torch.arange(S, device=dev), dim=-1))


def test_local_chunk_attention_properties_at_c4(dev):
    """(8, 8192, 512) local attention: with V = 1 every output is the softmax mass 1; a perturbed key at position t changes only
    the queries that can see it -- t .. end of the NEXT chunk -- and nothing else, bit for bit"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(1)
    B, T, H, dh = 8, RT, RH, 64
    d = H * dh
    q = torch.randn(B, T, d, device=dev).bfloat16()
    k = torch.randn(B, T, d, device=dev).bfloat16()
    ones = torch.ones(B, T, d, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B, 1, T, d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, 1, H, T, device=dev)
    ops.chunk_attn_fwd(q, k, ones, None, out, lse, B, T, H, dh, 1, 0, T * d, d)
    assert (out.float() - 1.0).abs().max().item() < 1e-2
    v = torch.randn(B, T, d, device=dev).bfloat16()
    ops.chunk_attn_fwd(q, k, v, None, out, lse, B, T, H, dh, 1, 0, T * d, d)
    base, base_lse = out.clone(), lse.clone()
    t = 5000                                                       # chunk 78 = [4992, 5056); next chunk ends at 5120
    k2 = k.clone(); k2[:, t] = k2[:, t] * -1.5
    ops.chunk_attn_fwd(q, k2, v, None, out, lse, B, T, H, dh, 1, 0, T * d, d)
    lo, hi = t, (t // 64 + 2) * 64
    assert torch.equal(out[:, :, :lo], base[:, :, :lo]) and torch.equal(out[:, :, hi:], base[:, :, hi:])
    assert torch.equal(lse[..., :lo], base_lse[..., :lo]) and torch.equal(lse[..., hi:], base_lse[..., hi:])
    changed = (out[:, :, lo:hi] != base[:, :, lo:hi]).any(-1).float().mean().item()
    assert changed > 0.9


def test_reformer_model_properties_at_c4(dev):
    """the C4 model (3 local + 3 LSH layers, T = 8192) with the hash rotations given explicitly: finite logits of the right
    shape, an initial loss near ln V, batch equivariance bit for bit; and the all-local variant is bit-exactly causal"""
    import math
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=RT, axial_pos_shape=(64, 128), num_hashes=1)
    m = MyReformerModelWithLMHead(cfg, device=dev, seed=3).eval()
    torch.manual_seed(2)
    ids = torch.randint(4, V, (2, RT), device=dev)
    rot = {l: torch.randn(RH, 64, 1, 16) for l, kind in enumerate(cfg.attn_layers) if kind == 'lsh'}     # buckets [16, 16]
    o = m(input_ids=ids, labels=ids, rotations=rot)
    lg = o.logits.float().clone()                   # the logits are a view of the engine's workspace: keep a copy
    assert lg.shape == (2, RT, V) and torch.isfinite(lg).all()
    assert abs(o.loss.item() - math.log(V)) < 0.5
    o_sw = m(input_ids=ids.flip(0), rotations=rot)
    assert torch.equal(o_sw.logits.float().flip(0), lg)
    # every LSH layer really sorted 8192 keys into 256 buckets
    for l in rot:
        b = m.engine.last_buckets[l]
        assert b.min().item() >= 0 and b.max().item() < 256 and b.unique().numel() > 200
    del m, o, o_sw
    cfg_l = MyReformerConfig('small', vocab_size=V, max_position_embeddings=RT, axial_pos_shape=(64, 128), num_hashes=1,
                             attn_layers=['local'] * 6)
    ml = MyReformerModelWithLMHead(cfg_l, device=dev, seed=3).eval()
    a = ml(input_ids=ids).logits.float().clone()
    t = 6000
    ids2 = ids.clone(); ids2[:, t] = (ids2[:, t] + 11) % (V - 4) + 4
    b2 = ml(input_ids=ids2).logits.float()
    assert torch.equal(a[:, :t], b2[:, :t]) and not torch.equal(a[:, t:], b2[:, t:])


def _ref_reformer(cfg, sd, num_buckets):
    from oracle.reformer_ref import RefReformerConfig, RefReformer, param_shapes
    rc = RefReformerConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads,
        attention_head_size=cfg.attention_head_size, feed_forward_size=cfg.feed_forward_size, attn_layers=list(cfg.attn_layers),
        max_position_embeddings=cfg.max_position_embeddings, axial_pos_shape=tuple(cfg.axial_pos_shape),
        axial_pos_embds_dim=tuple(cfg.axial_pos_embds_dim), num_hashes=cfg.num_hashes, chunk_length=cfg.lsh_attn_chunk_length,
        layer_norm_eps=cfg.layer_norm_eps, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    ref = RefReformer(rc, {k: sd[k].float() for k in param_shapes(rc)})
    ref.num_buckets = num_buckets
    return ref


@pytest.mark.parametrize('wscale', [1.0, 4.0])
def test_c4_reformer_forward_vs_oracle(dev, wscale):
    """SURVEY C4 (BASELINE.json configs[3]): Reformer 6L / 512d / H8 (3 local + 3 LSH layers), T = 8192, axial 64 x 128, one
    hash round, B = 1 -- HIP logits and loss against oracle/reformer_ref.py (pinned on the HF goldens) run on the host with the
    same bf16-representable weights and the same hash rotations.  A bucket is an arg-max over bf16 activations here and over fp32
    ones in the oracle, and one flipped token reroutes its attention (a discrete difference), so (i) the hashing is compared on
    its own -- the share of the 8 x 8192 bucket ids per LSH layer that agree -- and (ii) the logits are compared with the oracle
    given the bucket ids the HIP path used (HF's LSH attention accepts ready-made `buckets` for the same reason).  Two weight
    scales: the reference's init (logits O(0.3)) and 4x (logits O(5), the scale of the HF golden fixtures)."""
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=RT, axial_pos_shape=(64, 128), num_hashes=1)
    m = MyReformerModelWithLMHead(cfg, device=dev, seed=5).eval()
    sd = {k: (v * wscale if v.dim() > 1 and 'position_embeddings' not in k else v).to(torch.bfloat16).float()
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(4, V, (1, RT), generator=g)
    lab = ids.clone(); lab[0, RT - 300:] = -100
    rot = {l: torch.randn(RH, 64, 1, 16, generator=g) for l, kind in enumerate(cfg.attn_layers) if kind == 'lsh'}
    m.engine.keep_buckets = True
    with torch.no_grad():
        o = m(input_ids=ids.to(dev), labels=lab.to(dev), rotations=rot)
    lg = o.logits.float().cpu().clone()
    bk = {l: b.cpu().clone() for l, b in m.engine.last_buckets.items() if b is not None}
    assert sorted(bk) == sorted(rot)
    ref = _ref_reformer(cfg, sd, m.engine.num_buckets)
    with torch.no_grad():
        r_own, _ = ref.forward(ids, rotations=rot, labels=lab)                        # the oracle's own hashing
        own_bk = {l: ref.last_buckets[l].clone() for l in rot}
        r_lg, r_loss = ref.forward(ids, rotations=rot, labels=lab, buckets=bk)        # the HIP path's bucket assignment
    err = (lg - r_lg).abs()
    e_max, e_999, e_mean = err.max().item(), _quant(err, 0.999), err.mean().item()
    scale = r_lg.abs().mean().item()
    print(f'C4 reformer x{wscale}: |dlogit| max {e_max:.4f} p99.9 {e_999:.4f} mean {e_mean:.5f} (mean |logit| {scale:.3f}); '
          f'loss {o.loss.item():.5f} vs {r_loss.item():.5f}')
    first = min(rot)
    for l in rot:
        agree = (own_bk[l].reshape(-1) == bk[l].reshape(-1)).float().mean().item()
        print(f'   layer {l}: bucket agreement with the oracle\'s own hashing {agree:.4f}')
        # the first LSH layer sees inputs that differ by bf16 rounding only; deeper ones also see the attention the earlier
        # flips rerouted (in the oracle's own-hashing pass), which at 4x weights compounds: measured 0.986 / 0.882 / 0.813
        assert agree > (0.97 if (l == first or wscale == 1.0) else 0.75), (l, agree)
    # with its own (fp32) hashing the oracle differs from the HIP path only where a token changed bucket
    d_own = (lg - r_own).abs()
    print(f'   against the oracle with its own buckets: max {d_own.max().item():.4f} mean {d_own.mean().item():.5f}')
    # measured: x1 0.017 / 0.010 / 0.0025 at mean |logit| 0.51; x4 0.116 / 0.066 / 0.0155 at mean |logit| 2.1 (0.5-0.75 % in the mean)
    tol = {1.0: (3e-2, 2e-2, 5e-3), 4.0: (2e-1, 1e-1, 2.5e-2)}[wscale]
    assert e_max < tol[0] and e_999 < tol[1] and e_mean < tol[2]
    assert abs(o.loss.item() - r_loss.item()) / r_loss.item() < (1e-3 if wscale == 1.0 else 5e-3)
    top2 = r_lg.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol[1]
    assert (lg.argmax(-1) == r_lg.argmax(-1))[clear].all()


# ----------------------------------------------------------------------------------------------------------------------
# Full-size comparisons with the CPU oracle (oracle/transfoxl_ref.py run on the host): BASELINE.json configs[1] = SURVEY C2,
# configs[2] = C3 and configs[4] = C5; dropout 0, B = 1 on the oracle side.
#
# Tolerances (measured with scripts/diag_fullsize_error.py, which also prints the per-layer hidden-state error):
#  * at the reference's own init scale (`_init_weights`, std 0.02: the scale bench.py runs) the HIP log-probs differ from the fp32
#    oracle by max 0.024 (C2) / 0.033 (C3), mean 0.003 / 0.005, loss by < 1e-5 relative, arg-max identical everywhere;
#  * the SAME fp32 oracle with only its module outputs rounded to bf16 (fp32 arithmetic, bf16 storage between operators --
#    the precision class of the HIP path, which additionally rounds the MFMA operands q + bias, P and the attention output)
#    differs from itself by max 0.016 / 0.026, mean 0.0024 / 0.0037: the HIP path sits at 1.3-1.45x that envelope at every
#    quantile and every layer;
#  * at 3x the init scale a 12-layer model is chaotic in bf16 (envelope itself: mean 0.033, max 0.92 at C3), so the 3x stress
#    case is asserted at C2 only and RELATIVE to the envelope measured in the same test.
# ----------------------------------------------------------------------------------------------------------------------
def _oracle_pair(dev, preset, n_layer, T_, M_, seed, wscale=1.0):
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    torch.manual_seed(seed)
    kw = dict(vocab_size=V, n_layer=n_layer, mem_len=M_, max_length=T_, cutoffs=[], dropout=0.0)
    ref = RefTransfoXLLMHeadModel(RefXLConfig.from_preset(preset, **kw))
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() > 1 and 'layer_norm' not in n:
                p.mul_(wscale)
            p.copy_(p.to(torch.bfloat16).float())          # bf16-representable: both sides hold the same weights
    m = MyTransfoXLLMHeadModel(MyTransfoXLConfig(preset, **kw), device=dev)
    m.load_state_dict(ref.state_dict())
    return ref.eval(), m.eval()


def _bf16_storage_forward(ref, ids, lab):
    """the fp32 oracle with every module output (Linear, LayerNorm, Embedding, positional table) rounded to bf16 as it is written"""
    from torch import nn
    from oracle import transfoxl_ref as R
    kinds = (nn.Linear, nn.LayerNorm, nn.Embedding, R.PositionalEmbedding)
    rnd = lambda mod, inp, out: out.to(torch.bfloat16).float() if torch.is_tensor(out) else out
    hooks = [mod.register_forward_hook(rnd) for mod in ref.modules() if isinstance(mod, kinds)]
    try:
        with torch.no_grad():
            return ref(ids, labels=lab)
    finally:
        for h in hooks:
            h.remove()


def _quant(x, k):
    s = x.flatten().sort().values
    return s[min(int(k * s.numel()), s.numel() - 1)].item()


def _fwd_vs_oracle(dev, preset, n_layer, T_, M_, seed, wscale, tol_max, tol_p999, tol_mean, envelope=None):
    ref, m = _oracle_pair(dev, preset, n_layer, T_, M_, seed, wscale)
    g = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(4, V, (1, T_), generator=g)
    lab = ids.clone(); lab[0, T_ - 100:] = -100                   # a padded tail, as the collator produces
    with torch.no_grad():
        ro = ref(ids, labels=lab)
        o = m(input_ids=ids.to(dev), labels=lab.to(dev))
    lp, rlp = o.prediction_scores.float().cpu(), ro.prediction_scores
    assert lp.shape == rlp.shape == (1, T_, V)
    err = (lp - rlp).abs()
    e_max, e_999, e_mean = err.max().item(), _quant(err, 0.999), err.mean().item()
    print(f'{preset} {n_layer}L T={T_} x{wscale}: |dlogp| max {e_max:.4f} p99.9 {e_999:.4f} mean {e_mean:.5f}; '
          f'loss {o.loss.item():.5f} vs {ro.loss.item():.5f}')
    if envelope is not None:        # relative to the bf16-storage form of the same oracle on the same inputs
        env = (_bf16_storage_forward(ref, ids, lab).prediction_scores - rlp).abs()
        v_999, v_mean = _quant(env, 0.999), env.mean().item()
        print(f'   bf16-storage envelope of the oracle itself: max {env.max().item():.4f} p99.9 {v_999:.4f} mean {v_mean:.5f}')
        assert e_mean < envelope * v_mean and e_999 < envelope * v_999
        # hidden states per layer (the mems are the layer inputs): same ratio, no layer stands out
        for l in range(1, n_layer):
            hr = ro.mems[l][:, 0]
            rel = ((o.mems[l][:, 0].float().cpu() - hr).norm() / hr.norm()).item()
            assert rel < 2.5e-2, (l, rel)
    assert e_max < tol_max and e_999 < tol_p999 and e_mean < tol_mean
    assert abs(o.loss.item() - ro.loss.item()) / ro.loss.item() < (1e-3 if wscale == 1.0 else 1e-2)
    a, b = o.losses.float().cpu().flatten().sort().values, ro.losses.flatten().sort().values
    assert (a - b).abs().max().item() < tol_max
    # arg-max agreement wherever the oracle's top-2 margin exceeds twice the p99.9 tolerance
    top2 = rlp.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol_p999
    assert (lp.argmax(-1) == rlp.argmax(-1))[clear].all() and clear.float().mean().item() > 0.2
    return m


def test_c2_forward_vs_oracle(dev):
    """6L / 512d / H8, T = M = 1024 -- HIP log-probs, per-token NLLs and loss vs the fp32 oracle at the reference's
    init scale (absolute tolerances) and at 3x that scale (relative to the bf16-storage envelope); then the C3-style properties at
    this shape (normalisation, causality, batch equivariance, segmentation invariance)"""
    T2 = 1024
    _fwd_vs_oracle(dev, 'small', 6, T2, T2, seed=21, wscale=1.0, tol_max=4e-2, tol_p999=2.5e-2, tol_mean=6e-3)
    m = _fwd_vs_oracle(dev, 'small', 6, T2, T2, seed=21, wscale=3.0, tol_max=2e-1, tol_p999=8e-2, tol_mean=1.6e-2, envelope=1.6)
    torch.manual_seed(2)
    ids = torch.randint(4, V, (3, T2), device=dev)
    lp = m(input_ids=ids).prediction_scores.float()
    assert (lp.exp().sum(-1) - 1.0).abs().max().item() < 2e-3
    t = 700
    ids2 = ids.clone(); ids2[:, t] = (ids2[:, t] + 7) % (V - 4) + 4
    lp2 = m(input_ids=ids2).prediction_scores.float()
    assert torch.equal(lp2[:, :t], lp[:, :t]) and not torch.equal(lp2[:, t:], lp[:, t:])
    assert torch.equal(m(input_ids=ids.flip(0)).prediction_scores.float().flip(0), lp)
    o1 = m(input_ids=ids[:, :512]); o2 = m(input_ids=ids[:, 512:], mems=o1.mems)
    seg = torch.cat([o1.prediction_scores, o2.prediction_scores], 1).float()
    assert (seg - lp).abs().max().item() < 1e-1 and (seg - lp).abs().mean().item() < 1e-2


def _grad_table(named_ref_grads, eng, limits, skip=()):
    bad, worst = {}, (0.0, 1.0)
    for name, rg in named_ref_grads:
        if name in skip:
            continue
        g = eng.g32(name).float().cpu().reshape(rg.shape)
        e = ((g - rg).norm() / (rg.norm() + 1e-12)).item()
        cos = torch.nn.functional.cosine_similarity(g.flatten(), rg.flatten(), dim=0).item()
        lim = limits(name)
        if name.split('.')[-2:] != ['r_net', 'weight']:
            worst = (max(worst[0], e), min(worst[1], cos))
        if e > lim[0] or cos < lim[1]:
            bad[name] = (round(e, 4), round(cos, 5))
    return bad, worst


def _rnet_limits(ref, ids, lab, eng, mems=None):
    """Limits of the r_net.weight gradients, RELATIVE to what bf16 storage of activations and gradient streams costs the oracle
    itself on the same inputs (`_bf16_storage_train_step`; call after the exact backward has been read): rel-Frobenius
    <= max(6 %, 1.5 x that envelope), never beyond 20 %; cosine likewise (1 - cosine is quadratic in the deviation).  Round 6
    measured why this tensor reads 2 - 18 % while the others read 1 - 6 %: the storage envelope of the oracle is itself 2 - 8.5 % on
    it (12-layer C3), the kernels sit at 1.07 - 1.46 x that like on every other tensor, and the layer-local rounding the kernels
    add (dS and q + r_r_bias in bf16 inside the dRd contraction) is 0.24 % per layer in isolation, 0.17 % with a hi + lo split
    of dS (scripts/exp_rnet_fidelity.py, profiles/r06_rnet_fidelity.txt)."""
    exact = {n: p.grad.clone() for n, p in ref.named_parameters() if n.endswith('r_net.weight')}
    env = _bf16_storage_train_step(ref, ids, lab, mems=mems)
    lims, table = {}, {}
    for n, rg in exact.items():
        ee = ((env[n] - rg).norm() / (rg.norm() + 1e-12)).item()
        ecos = torch.nn.functional.cosine_similarity(env[n].flatten(), rg.flatten(), dim=0).item()
        g = eng.g32(n).float().cpu().reshape(rg.shape)
        table[int(n.split('.')[2])] = (round(((g - rg).norm() / (rg.norm() + 1e-12)).item(), 4), round(ee, 4))
        lims[n] = (min(0.20, max(0.06, 1.5 * ee)), max(0.98, min(0.998, 1.0 - 2.25 * (1.0 - ecos))))
    print(f'r_net.weight per layer (HIP rel, bf16-storage oracle rel): {table}')
    return lims


def test_c2_train_step_gradients_vs_oracle(dev):
    """SURVEY C2 at full size, one training step's backward: 6L / 512d / H8, T = M = 1024, B = 1, mode R (no carried mems: the
    reference's training), dropout 0, a padded label tail -- loss and EVERY parameter's gradient against the fp32 oracle's
    autograd on the host.  Limits as at the small shapes (tests/test_xl_model_gpu.py): rel-Frobenius <= 6 %, cosine >= 0.998;
    r_net.weight relative to the bf16-storage envelope of the oracle itself (`_rnet_limits`; a fixed 20 % / 0.98 until round 6)."""
    T2 = 1024
    ref, m = _oracle_pair(dev, 'small', 6, T2, T2, seed=33, wscale=1.0)
    ref.train(); m.train()
    g = torch.Generator().manual_seed(34)
    ids = torch.randint(4, V, (1, T2), generator=g)
    lab = ids.clone(); lab[0, T2 - 100:] = -100
    ro = ref(ids, labels=lab)
    ro.loss.backward()
    m.zero_grad()
    o = m(input_ids=ids.to(dev), labels=lab.to(dev))
    m.backward()
    torch.cuda.synchronize()
    assert abs(o.loss.item() - ro.loss.item()) / ro.loss.item() < 1e-3
    named = [(n, p.grad.clone()) for n, p in ref.named_parameters()]
    rnet = _rnet_limits(ref, ids, lab, m.engine)
    lim = lambda k: rnet[k] if k.endswith('r_net.weight') else (0.06, 0.998)
    bad, worst = _grad_table(named, m.engine, lim, skip=('crit.out_layers.0.weight',))
    print(f'C2 gradients vs oracle: worst rel {worst[0]:.4f}, worst cosine {worst[1]:.5f} (r_net.weight aside)')
    assert not bad, bad


@pytest.mark.parametrize('with_mem', [False, True])
def test_c3_shape_two_layer_gradients_vs_oracle(dev, with_mem):
    """The C3 layer shapes exactly as the bench runs them (768d / H12 / dh64 / F3072, T = M = 2048, V = 1190) in a TWO-layer model,
    so that the oracle's autograd fits the host comfortably (the 12-layer step needs ~25 GB): loss and every parameter's gradient,
    B = 1, (a) mode R -- no carried mems, the reference's training, half of every window is phantom distances -- and (b) with
    2048 carried memory rows per layer (Kc = 4096: the attention kernels' other regime).  Same limits as the small shapes, with
    one documented exception: an untrained model with tied embeddings predicts its own INPUT token with p near 1 while the label
    is the next token, so the column sums over tokens of the logit gradient (sum_t p_t[v] - count_t[label = v]) are differences of
    nearly equal numbers.  The output bias gradient is such a sum -- exact here because the logit gradient is carried as a two-term
    bf16 sum (mxl_adaptive_nll_bwd_split; with one term it was 20 % off) -- and so is the last LayerNorm's bias gradient,
    sum_t dh_t, whose terms are rounded to bf16 per token: 7 % / cosine 0.997 measured, limit 10 % / 0.995."""
    ref, m = _oracle_pair(dev, 'base', 2, T, M, seed=41, wscale=1.0)
    ref.train(); m.train()
    g = torch.Generator().manual_seed(42)
    ids = torch.randint(4, V, (1, T), generator=g)
    lab = ids.clone(); lab[0, T - 100:] = -100
    mems_r = mems_h = None
    if with_mem:
        mems_r = [(torch.randn(M, 1, D, generator=g) * 0.5).to(torch.bfloat16).float() for _ in range(2)]
        mems_h = [x.to(dev) for x in mems_r]
    ro = ref(ids, labels=lab, mems=mems_r)
    ro.loss.backward()
    m.zero_grad()
    o = m(input_ids=ids.to(dev), labels=lab.to(dev), mems=mems_h)
    m.backward()
    torch.cuda.synchronize()
    assert abs(o.loss.item() - ro.loss.item()) / ro.loss.item() < 1e-3
    last_ln_bias = 'transformer.layers.1.pos_ff.layer_norm.bias'
    named = {n: p.grad.clone() for n, p in ref.named_parameters()}
    rnet = _rnet_limits(ref, ids, lab, m.engine, mems=mems_r)
    lim = lambda k: rnet[k] if k.endswith('r_net.weight') else (0.10, 0.995) if k == last_ln_bias else (0.06, 0.998)
    bad, worst = _grad_table([(n, g_) for n, g_ in named.items() if n != last_ln_bias], m.engine, lim,
                             skip=('crit.out_layers.0.weight',))
    bad2, w2 = _grad_table([(last_ln_bias, named[last_ln_bias])], m.engine, lim)
    bad.update(bad2)
    print(f'C3-shape 2-layer gradients vs oracle (mems: {with_mem}): worst rel {worst[0]:.4f}, worst cosine {worst[1]:.5f}; '
          f'last LayerNorm bias {w2[0]:.4f} / {w2[1]:.5f}')
    assert not bad, bad


@pytest.mark.parametrize('wscale', [1.0, 4.0])
def test_c4_reformer_train_step_gradients_vs_oracle(dev, wscale):
    """SURVEY C4 at full size, one training step's backward: Reformer 6L / 512d (3 local + 3 LSH), T = 8192, B = 1, dropout 0 --
    loss and every parameter's gradient against the pinned oracle's autograd, the oracle given the bucket assignment the HIP path
    used (see test_c4_reformer_forward_vs_oracle).  At the reference's init scale the limits are those of the small shapes
    (6 % rel-Frobenius / cosine 0.998); at 4x the init (the scale of the HF fixtures) 128 chunks deep in a sequence the bf16
    activations cost a little more: 9 % / 0.996 (measured 7.5 % / 0.9972 at worst: word embeddings and layer 0's q / k)."""
    from oracle.reformer_ref import param_shapes
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=RT, axial_pos_shape=(64, 128), num_hashes=1,
                           hidden_dropout_prob=0.0, local_attention_probs_dropout_prob=0.0)
    m = MyReformerModelWithLMHead(cfg, device=dev, seed=7).train()
    sd = {k: (v * wscale if v.dim() > 1 and 'position_embeddings' not in k else v).to(torch.bfloat16).float()
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(4, V, (1, RT), generator=g)
    lab = ids.clone(); lab[0, RT - 300:] = -100
    rot = {l: torch.randn(RH, 64, 1, 16, generator=g) for l, kind in enumerate(cfg.attn_layers) if kind == 'lsh'}
    m.engine.keep_buckets = True
    m.zero_grad()
    o = m(input_ids=ids.to(dev), labels=lab.to(dev), rotations=rot)
    m.backward()
    torch.cuda.synchronize()
    bk = {l: b.cpu().clone() for l, b in m.engine.last_buckets.items() if b is not None}
    ref = _ref_reformer(cfg, sd, m.engine.num_buckets)
    for v in ref.p.values():
        v.requires_grad_(True)
    _, r_loss = ref.forward(ids, rotations=rot, labels=lab, buckets=bk)
    r_loss.backward()
    assert abs(o.loss.item() - r_loss.item()) / r_loss.item() < 5e-3
    names = list(param_shapes(ref.c))
    lim = (0.06, 0.998) if wscale == 1.0 else (0.09, 0.996)
    bad, worst = _grad_table([(k, ref.p[k].grad) for k in names if ref.p[k].grad is not None], m.engine, lambda k: lim)
    print(f'C4 reformer x{wscale} gradients vs oracle: worst rel {worst[0]:.4f}, worst cosine {worst[1]:.5f}')
    assert not bad, bad


def test_c3_forward_vs_oracle(dev):
    """SURVEY C3 (the headline config): 12L / 768d / H12, T = M = 2048, B = 1, the reference's init scale -- HIP forward vs the
    fp32 oracle (absolute tolerances; the 3x stress scale is asserted at C2, see the comment block above)"""
    _fwd_vs_oracle(dev, 'base', 12, T, M, seed=23, wscale=1.0, tol_max=5e-2, tol_p999=3e-2, tol_mean=8e-3)


def test_c5_decode_batch64_ring_wrap(dev):
    """12L / 768d, M = 2048, B = 64 prompts x 256 tokens, greedy, one hipGraph replay per token, 1920 generated tokens
    (positions up to 2176: the K/V rings wrap at 2048), reference init scale.  (i) every generated token is the arg-max of the
    step's own log-probs; (ii) teacher-forcing the decoded ids through one-shot HIP forwards (two 1088-token segments with
    carried mems) reproduces the per-step log-probs (segmentation invariance at size: max |dlogp| < 5e-2) and the greedy choice
    wherever the one-shot top-2 margin is clear; (iii) two of the rows against the CPU oracle's HF-style greedy loop for the
    first 32 tokens (token for token; a fork is accepted only at a near-tie of the oracle's own log-probs)."""
    from symbolic_music_generation_amd.generate import XLDecoder
    B, Tp, TOT = 64, 256, 2176
    ref, m = _oracle_pair(dev, 'base', 12, 2048, 2048, seed=29, wscale=1.0)
    g = torch.Generator().manual_seed(31)
    prompt = torch.randint(4, V, (B, Tp), generator=g)
    dec = XLDecoder(m.engine, B, TOT, seed=1)
    dec.trace = torch.zeros(B, TOT, V, device=dev)
    with torch.no_grad():
        ids = dec.generate(prompt.to(dev), TOT, do_sample=False, use_graph=True)
    assert ids.shape == (B, TOT) and torch.equal(ids[:, :Tp].cpu(), prompt)
    tr = dec.trace[:, Tp - 1:TOT - 1]                              # log-probs that chose tokens Tp .. TOT-1
    chosen = tr.gather(-1, ids[:, Tp:].unsqueeze(-1)).squeeze(-1)
    assert torch.equal(chosen, tr.max(-1).values)                  # (i) greedy = arg-max of the step's log-probs
    assert (tr.exp().sum(-1) - 1).abs().max().item() < 2e-3
    # (ii) one-shot forwards on the decoded ids
    with torch.no_grad():
        o1 = m(input_ids=ids[:, :1088])
        lp1 = o1.prediction_scores[:, Tp - 1:].float().clone()
        o2 = m(input_ids=ids[:, 1088:], mems=o1.mems)
        lp2 = o2.prediction_scores[:, :TOT - 1 - 1088].float()
    one = torch.cat([lp1, lp2], 1)
    err = (one - tr).abs()
    post = err[:, 2048 - Tp:]                                       # steps taken after the rings wrapped
    print(f'C5 decode vs one-shot: max |dlogp| {err.max().item():.4f} mean {err.mean().item():.5f}; after the wrap: max '
          f'{post.max().item():.4f} mean {post.mean().item():.5f}')
    assert err.max().item() < 5e-2 and err.mean().item() < 5e-3
    top2 = one.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 5e-2
    agree = one.argmax(-1) == ids[:, Tp:]
    assert agree[clear].all() and agree.float().mean().item() > 0.97, agree.float().mean().item()
    assert int(dec.t_dev.item()) >= 2048 + 100                      # the wrap really happened
    # (iii) oracle greedy loop (HF style: prompt, then one token at a time with carried mems) on two rows
    rows = [0, B - 1]
    want = ref.greedy_generate(prompt[rows], max_length=Tp + 32)
    got = ids[rows, :Tp + 32].cpu()
    mism = (got != want).nonzero()
    print(f'C5 greedy vs oracle: {int((got == want)[:, Tp:].sum())} of {2 * 32} generated tokens identical'
          + (f', first fork at {mism[0].tolist()}' if mism.numel() else ''))
    if mism.numel():            # a fork is legitimate only at a bf16 near-tie of the oracle's own log-probs
        r0, t0 = mism[0].tolist()
        with torch.no_grad():
            rl = ref(want[r0:r0 + 1, :t0]).prediction_scores[0, -1]
        margin = (rl.topk(2).values[0] - rl.topk(2).values[1]).item()
        assert margin < 5e-2, f'greedy decode diverges from the oracle at row {r0} position {t0} with margin {margin}'


# ---------------------------------------------------------------------------------------------------------------------------
# round 3: the headline config's own backward, sampler and dropout at size
# ---------------------------------------------------------------------------------------------------------------------------
class _RoundBoth(torch.autograd.Function):
    """identity whose value AND gradient are rounded to bf16: one storage rounding in each direction"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


import contextlib


@contextlib.contextmanager
def bf16_storage(ref):
    """the fp32 oracle with every module output (Linear, LayerNorm, Embedding, positional table) rounded to bf16 as it is written
    and every gradient rounded to bf16 as it flows back through the same points: what bf16 STORAGE of activations and gradient
    streams costs by itself, with exact arithmetic everywhere else"""
    from torch import nn
    from oracle import transfoxl_ref as R
    kinds = (nn.Linear, nn.LayerNorm, nn.Embedding, R.PositionalEmbedding)
    rnd = lambda mod, inp, out: _RoundBoth.apply(out) if torch.is_tensor(out) else out
    hooks = [mod.register_forward_hook(rnd) for mod in ref.modules() if isinstance(mod, kinds)]
    try:
        yield ref
    finally:
        for h in hooks:
            h.remove()


def _bf16_storage_train_step(ref, ids, lab, mems=None):
    """one backward of the oracle under `bf16_storage`.  Returns {name: gradient}."""
    with bf16_storage(ref):
        ref.zero_grad()
        ref(ids, labels=lab, mems=mems).loss.backward()
        return {n: p.grad.clone() for n, p in ref.named_parameters()}


def test_c3_train_step_gradients_vs_oracle(dev):
    """SURVEY C3 at FULL depth: 12L / 768d / H12 / dh64 / F3072, T = M = 2048, V = 1190, B = 1, mode R (fresh zero mems: the
    reference's training, musicnlp/models/transformer_xl.py:130-221 under HF Trainer), dropout 0, a padded label tail -- loss and
    EVERY parameter's gradient of the 12-layer backward against the fp32 oracle's autograd.  The oracle recomputes each layer in
    its backward (`checkpoint_layers`: the same arithmetic; the dense (2048, 4096, 12) score tensors of twelve layers would need
    ~25 GB otherwise).
    Limits: the fixed ones of C2 and of the two-layer C3 test (rel-Frobenius <= 6 %, cosine >= 0.998; r_net.weight <= 7 % /
    >= 0.995, the last LayerNorm's bias <= 10 % / >= 0.995) or, where bf16 STORAGE alone costs more than that, 1.5 x the deviation
    of the same oracle run with bf16 storage of activations and gradient streams and exact arithmetic everywhere else
    (`_bf16_storage_train_step`), never beyond 15 % / 0.985.  Measured: that envelope is 5.9 % (layer 11) to 8.8 % (layer 0) on
    the worst tensor of a layer (the first FFN weight: 2048 tokens, one sequence), the HIP path 1.19 x it at every depth --
    i.e. the 12-layer gradient error is what the storage format costs, not something the kernels add with depth."""
    ref, m = _oracle_pair(dev, 'base', 12, T, M, seed=51, wscale=1.0)
    ref.train(); m.train()
    ref.transformer.checkpoint_layers = True
    g = torch.Generator().manual_seed(52)
    ids = torch.randint(4, V, (1, T), generator=g)
    lab = ids.clone(); lab[0, T - 100:] = -100
    ro = ref(ids, labels=lab)
    ro.loss.backward()
    exact = {n: p.grad.clone() for n, p in ref.named_parameters()}
    m.zero_grad()
    o = m(input_ids=ids.to(dev), labels=lab.to(dev))
    m.backward()
    torch.cuda.synchronize()
    assert abs(o.loss.item() - ro.loss.item()) / ro.loss.item() < 1e-3
    env = _bf16_storage_train_step(ref, ids, lab)
    last_ln_bias = 'transformer.layers.11.pos_ff.layer_norm.bias'

    def err(a, b):
        return (((a - b).norm() / (b.norm() + 1e-12)).item(),
                torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item())

    bad, per_layer = {}, {}
    for name, rg in exact.items():
        if name == 'crit.out_layers.0.weight':
            continue
        e, cos = err(m.engine.g32(name).float().cpu().reshape(rg.shape), rg)
        ee, ecos = err(env[name], rg)
        parts = name.split('.')
        layer = int(parts[2]) if parts[1] == 'layers' else None
        # r_r_bias: like r_net.weight a sum of the un-skewed score gradient over every (query, distance) cell of a head, with
        # cancellation (the rows of dG sum to zero): 6.2 % measured at layer 1, the others 2-5 %
        # r_net.weight: measured 2.2 - 10.0 % per layer, the bf16-storage oracle 2.1 - 8.5 % on the same tensor: 7 % or 1.5 x that envelope
        # (a fixed 12 % until round 6, 20 % until round 4; see _rnet_limits)
        base = ((0.07, 0.995) if name.endswith('r_net.weight') else (0.10, 0.995) if name == last_ln_bias or name.endswith('r_r_bias')
                else (0.06, 0.998))
        # "1.5 x the deviation" on both measures: 1 - cosine is quadratic in the relative deviation, so 1.5 x in deviation is
        # 2.25 x in 1 - cosine.  (Until round 4 the cosine term used 1.5 x, i.e. 1.22 x in deviation, while the HIP path sits at
        # 1.19 - 1.24 x the envelope on every tensor: two LayerNorm weights then read 0.99797 against a limit of 0.998.)
        lim = (min(0.15, max(base[0], 1.5 * ee)), max(0.985, min(base[1], 1.0 - 2.25 * (1.0 - ecos))))
        if e > lim[0] or cos < lim[1]:
            bad[name] = (round(e, 4), round(cos, 5), 'envelope', round(ee, 4), round(ecos, 5))
        if layer is not None and not name.endswith('r_net.weight'):
            w = per_layer.setdefault(layer, [0.0, 1.0, 0.0, 1.0])
            w[0], w[1], w[2], w[3] = max(w[0], e), min(w[1], cos), max(w[2], ee), min(w[3], ecos)
    for l in sorted(per_layer):
        w = per_layer[l]
        print(f'C3 12-layer gradients, layer {l:2d}: HIP worst rel {w[0]:.4f} cos {w[1]:.5f} | bf16-storage oracle {w[2]:.4f} / {w[3]:.5f}')
    rnet = {int(n.split('.')[2]): round(err(m.engine.g32(n).float().cpu().reshape(gr.shape), gr)[0], 4)
            for n, gr in exact.items() if n.endswith('r_net.weight')}
    rnet_env = {int(n.split('.')[2]): round(err(env[n], gr)[0], 4) for n, gr in exact.items() if n.endswith('r_net.weight')}
    print(f'r_net.weight rel per layer {rnet}')
    print(f'r_net.weight, bf16-storage oracle per layer {rnet_env}')
    assert not bad, bad


def test_c5_topk_sampling_in_graph_at_batch64(dev):
    """SURVEY C5 as bench.py times it: 12L / 768d, M = 2048, 64 prompts x 256 tokens, `top_k = 8` multinomial sampling
    (musicnlp/trainer/eval.py:277-326, README top_k 8) inside the hipGraph-captured step, generated to T = 2048.  (i) every
    sampled id lies in the top-8 of the log-probs of ITS OWN step; (ii) over the 64 x 1792 draws the number of times the k-th
    most likely token was taken matches its expectation sum p_k under the renormalised top-8 distribution (|z| < 5 per rank:
    the draws are independent given the log-probs) and the mean log-probability of the taken token matches its expectation;
    (iii) rows differ from each other and from the greedy continuation (the per-row RNG streams are distinct)."""
    from symbolic_music_generation_amd.generate import XLDecoder
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    B, Tp, TOT, K = 64, 256, 2048, 8
    cfg = MyTransfoXLConfig('base', max_length=2048, vocab_size=V, mem_len=2048, cutoffs=[])
    m = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).eval()
    with torch.no_grad():        # sharpen the untrained model a little so the top-8 probabilities are not all equal
        m.engine.P.mul_(2.0)
        m.engine.sync_weights() if hasattr(m.engine, 'sync_weights') else None
    g = torch.Generator().manual_seed(32)
    prompt = torch.randint(4, V, (B, Tp), generator=g)
    dec = XLDecoder(m.engine, B, TOT, seed=5)
    dec.trace = torch.zeros(B, TOT, V, device=dev)
    with torch.no_grad():
        ids = dec.generate(prompt.to(dev), TOT, do_sample=True, top_k=K, temperature=1.0, use_graph=True)
    assert ids.shape == (B, TOT) and torch.equal(ids[:, :Tp].cpu(), prompt)
    tr = dec.trace[:, Tp - 1:TOT - 1].double()                      # log-probs that chose tokens Tp .. TOT-1
    new = ids[:, Tp:]
    top = tr.topk(K, -1)
    hit = top.indices == new.unsqueeze(-1)
    assert hit.any(-1).all(), 'a sampled id outside the top-8 of its own step'                        # (i)
    p = torch.softmax(top.values, -1)                                # renormalised top-8 (HF: top-k filter, then renormalise)
    obs = hit.double().sum((0, 1))
    exp = p.sum((0, 1))
    var = (p * (1 - p)).sum((0, 1))
    z = (obs - exp) / var.sqrt()
    lp_taken = (torch.log(p) * hit).sum(-1)
    lp_exp = (p * torch.log(p)).sum(-1)
    lp_var = (p * torch.log(p) ** 2).sum(-1) - lp_exp ** 2
    z_lp = ((lp_taken - lp_exp).sum() / lp_var.sum().sqrt()).item()
    print(f'C5 top-8 sampling, {new.numel()} draws: rank counts {obs.long().tolist()} expected {[round(x) for x in exp.tolist()]} '
          f'z {[round(x, 2) for x in z.tolist()]}; taken-token log-prob z = {z_lp:.2f}; mean p(top-1) {p[..., 0].mean().item():.3f}')
    assert z.abs().max().item() < 5.0 and abs(z_lp) < 5.0                                             # (ii)
    assert p[..., 0].mean().item() < 0.9                             # the test has power: the distributions are not one-hot
    assert len({tuple(r.tolist()) for r in new[:, :16].cpu()}) > B // 2                               # (iii)
    assert (new != tr.argmax(-1)).float().mean().item() > 0.2


def test_c3_bench_batch_dropout_step_is_a_function_of_seed_and_step(dev):
    """bench.py's timed workload itself (C3, per-GPU batch 64, dropout 0.1 on): the masks are never stored, the backward
    REGENERATES them from (seed, step, site, element index).  (a) op level, at the step's own sizes (131072 x 768 and its 64-bit
    element indices): the mask `mxl_ln_residual_bwd` regenerates equals the one `mxl_ln_residual_fwd` applied, element for
    element, and another site / seed gives another mask; (b) whole step: run twice from the same parameters and the same
    rng_step -- same loss to fp32 atomic ordering, every parameter's gradient equal to 1e-4 rel-Frobenius (the weight-gradient /
    dRd / bias atomics reorder, nothing else may move); a different rng_step moves the loss."""
    from symbolic_music_generation_amd import ops
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    B = 64
    N = B * T
    # (a)
    x = torch.ones(N, D, device=dev, dtype=torch.bfloat16)
    res = torch.zeros_like(x)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, z = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.ln_residual_fwd(x, res, gam, bet, y, z, mean, rstd, drop_p=0.1, seed=1234567, site=9)
    keep_f = z != 0
    frac = keep_f.float().mean().item()
    assert abs(frac - 0.9) < 1e-3, frac
    dy = torch.ones_like(x)
    # a z with row variance (LayerNorm backward of a constant row is zero): the mask does not depend on z
    torch.manual_seed(0)
    z2 = torch.randn(N, D, device=dev).to(torch.bfloat16)
    mean2 = z2.float().mean(-1); rstd2 = (z2.float().var(-1, unbiased=False) + 1e-5).rsqrt()
    dy2 = torch.randn(N, D, device=dev).to(torch.bfloat16)
    dres, dx = torch.empty_like(x), torch.empty_like(x)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ops.ln_residual_bwd(dy2, None, z2, mean2, rstd2, gam, dres, dx, dg, db, drop_p=0.1, seed=1234567, site=9)
    nz = dres != 0
    assert torch.equal((dx != 0)[nz], keep_f[nz]) and nz.float().mean().item() > 0.99
    ops.ln_residual_bwd(dy2, None, z2, mean2, rstd2, gam, dres, dx, dg, db, drop_p=0.1, seed=1234567, site=10)
    assert ((dx != 0)[nz] != keep_f[nz]).float().mean().item() > 0.1
    del x, res, y, z, dy, z2, dy2, dres, dx, keep_f, nz
    torch.cuda.empty_cache()
    # (b)
    cfg = MyTransfoXLConfig('base', max_length=T, vocab_size=V, n_layer=12, mem_len=M, cutoffs=[])
    assert cfg.dropout == 0.1
    m = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    eng = m.engine
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(4, V, (B, T), generator=g).to(dev)

    def step(rng_step):
        eng.rng_step = rng_step
        with torch.no_grad():
            eng.zero_grad()
            o = m(input_ids=ids, labels=ids)
            eng.backward()
        torch.cuda.synchronize()
        return o.loss.item(), eng.G.clone()

    l0, g0 = step(3)
    l1, g1 = step(3)
    l2, _ = step(4)
    assert abs(l0 - l1) < 1e-5 * abs(l0), (l0, l1)
    assert abs(l0 - l2) > 1e-6 * abs(l0), 'another rng_step must draw other masks'
    worst = 0.0
    for name in eng.layout.real_names():
        a, b = eng.layout.view(g0, name).double(), eng.layout.view(g1, name).double()
        worst = max(worst, ((a - b).norm() / (a.norm() + 1e-30)).item())
    print(f'C3 B=64 dropout step twice: loss {l0:.6f} / {l1:.6f} (other step {l2:.6f}), worst gradient rel difference {worst:.2e}')
    assert worst < 1e-4


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the three full-size comparisons that were still self-checks or rested on a crutch.
# ----------------------------------------------------------------------------------------------------------------------
def test_c3_forward_with_carried_mems_vs_oracle(dev):
    """SURVEY C3 in mode S (`musicnlp/models/transformer_xl.py:223-241`: generation and segment recurrence hand `mems` back in):
    the second 2048-token segment of a stream with the first segment's memories carried in, 12L / 768d, B = 1, against the fp32
    oracle fed ITS OWN carried memories -- the whole recurrence end to end, not two HIP passes compared with each other.  The
    error budget is C3's one-segment budget plus what the first segment's hidden states (the memories) already carry."""
    ref, m = _oracle_pair(dev, 'base', 12, T, M, seed=41, wscale=1.0)
    g = torch.Generator().manual_seed(42)
    ids0 = torch.randint(4, V, (1, T), generator=g)
    ids1 = torch.randint(4, V, (1, T), generator=g)
    lab = ids1.clone(); lab[0, T - 100:] = -100
    with torch.no_grad():
        r0 = ref(ids0)
        r1 = ref(ids1, mems=r0.mems, labels=lab)
        o0 = m(input_ids=ids0.to(dev))
        o1 = m(input_ids=ids1.to(dev), mems=o0.mems, labels=lab.to(dev))
    for l in range(1, 12):          # the carried memories themselves (layer inputs of segment 0)
        hr = r0.mems[l][:, 0]
        rel = ((o0.mems[l][:, 0].float().cpu() - hr).norm() / hr.norm()).item()
        assert rel < 2.5e-2, (l, rel)
    lp, rlp = o1.prediction_scores.float().cpu(), r1.prediction_scores
    err = (lp - rlp).abs()
    e_max, e_999, e_mean = err.max().item(), _quant(err, 0.999), err.mean().item()
    print(f'C3 mode S (carried mems): |dlogp| max {e_max:.4f} p99.9 {e_999:.4f} mean {e_mean:.5f}; '
          f'loss {o1.loss.item():.5f} vs {r1.loss.item():.5f}')
    assert e_max < 6e-2 and e_999 < 3.5e-2 and e_mean < 9e-3
    assert abs(o1.loss.item() - r1.loss.item()) / r1.loss.item() < 1e-3
    top2 = rlp.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 7e-2
    assert (lp.argmax(-1) == rlp.argmax(-1))[clear].all() and clear.float().mean().item() > 0.2
    for l in range(1, 12):          # the memories handed on to a third segment
        hr = r1.mems[l][:, 0]
        rel = ((o1.mems[l][:, 0].float().cpu() - hr).norm() / hr.norm()).item()
        assert rel < 2.5e-2, (l, rel)


def test_c4_reformer_forward_vs_oracle_hashing_for_itself(dev):
    """SURVEY C4 end to end WITHOUT handing the oracle the HIP path's buckets: the oracle hashes its fp32 activations itself, the
    HIP path its bf16 ones, and the logits are compared on the tokens whose bucket agrees in EVERY head of EVERY LSH layer (a token
    that changed bucket was rerouted: a discrete difference, counted and bounded separately).  Agreeing tokens still see a few
    rerouted neighbours in their chunks, so the bound is on the mean and the 99th percentile, not on the maximum."""
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=RT, axial_pos_shape=(64, 128), num_hashes=1)
    m = MyReformerModelWithLMHead(cfg, device=dev, seed=15).eval()
    sd = {k: v.to(torch.bfloat16).float() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(16)
    ids = torch.randint(4, V, (1, RT), generator=g)
    rot = {l: torch.randn(RH, 64, 1, 16, generator=g) for l, kind in enumerate(cfg.attn_layers) if kind == 'lsh'}
    m.engine.keep_buckets = True
    with torch.no_grad():
        lg = m(input_ids=ids.to(dev), rotations=rot).logits.float().cpu().clone()
    bk = {l: b.cpu().clone() for l, b in m.engine.last_buckets.items() if b is not None}
    ref = _ref_reformer(cfg, sd, m.engine.num_buckets)
    with torch.no_grad():
        r_own, _ = ref.forward(ids, rotations=rot)
    same = torch.ones(RT, dtype=torch.bool)
    for l in rot:
        a, b_ = ref.last_buckets[l].reshape(-1, RT), bk[l].reshape(-1, RT)          # (heads x rounds, T)
        same &= (a == b_).all(0)
    frac = same.float().mean().item()
    err = (lg - r_own).abs()[0]                                                         # (T, V)
    e_same, e_diff = err[same], err[~same]
    print(f'C4, oracle hashing for itself: {frac:.4f} of the tokens keep their bucket in all {len(rot)} LSH layers x {RH} heads; '
          f'|dlogit| on them mean {e_same.mean().item():.5f} p99 {_quant(e_same, 0.99):.4f} max {e_same.max().item():.4f}; '
          f'on the rerouted ones mean {e_diff.mean().item() if e_diff.numel() else 0.0:.5f}')
    assert frac > 0.65               # 24 arg-max decisions per token at ~99 % agreement each
    # measured: 0.756 of the tokens agree; on them mean 0.0081, p99 0.043 (mean |logit| 0.5: 1.6 %) -- three times the error with the
    # buckets handed over (0.0025), because an agreeing token still attends to chunks whose other members were rerouted -- and half
    # the error of the rerouted tokens themselves (0.0164)
    assert e_same.mean().item() < 1.2e-2 and _quant(e_same, 0.99) < 6.5e-2
    assert e_diff.numel() == 0 or e_same.mean().item() < e_diff.mean().item()
    top2 = r_own.topk(2, -1).values[0]
    clear = ((top2[:, 0] - top2[:, 1]) > 0.3) & same
    assert (lg[0].argmax(-1) == r_own[0].argmax(-1))[clear].all()


def test_reformer_base_two_hash_rounds_at_seq4096_vs_oracle(dev):
    """The reference's own logged Reformer configuration (`musicnlp/models/reformer.py:30-43`, preset `base`: d = 768, 12 heads,
    num_hashes = 2; notebook/train/reformer.ipynb: seq 4096, axial 64 x 64, auto num_buckets = 128) at full width and length, two
    layers (one local, one LSH): logits and loss against the pinned oracle given the HIP path's bucket assignment, and the share
    of bucket ids that agree with the oracle's own hashing per round.  num_hashes = 2 was only covered at toy size before."""
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    Tr = 4096
    cfg = MyReformerConfig('base', vocab_size=420, max_position_embeddings=Tr, axial_pos_shape=(64, 64), attn_layers=['local', 'lsh'])
    assert cfg.num_hashes == 2 and cfg.hidden_size == 768 and cfg.num_attention_heads == 12
    m = MyReformerModelWithLMHead(cfg, device=dev, seed=25).eval()
    sd = {k: v.to(torch.bfloat16).float() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(26)
    ids = torch.randint(4, 420, (1, Tr), generator=g)
    lab = ids.clone(); lab[0, Tr - 200:] = -100
    rot = {1: torch.randn(12, 64, 2, 64, generator=g)}                  # (heads, dh, hash rounds, num_buckets / 2): 128 buckets, not factorised
    m.engine.keep_buckets = True
    with torch.no_grad():
        o = m(input_ids=ids.to(dev), labels=lab.to(dev), rotations=rot)
    assert m.engine.num_buckets == 128                                  # the notebook's logged value (SURVEY 8c fixture 3)
    bk = {l: b.cpu().clone() for l, b in m.engine.last_buckets.items() if b is not None}
    assert sorted(bk) == [1]
    lg = o.logits.float().cpu()
    ref = _ref_reformer(cfg, sd, m.engine.num_buckets)
    with torch.no_grad():
        ref.forward(ids, rotations=rot, labels=lab)
        own = ref.last_buckets[1].clone()
        r_lg, r_loss = ref.forward(ids, rotations=rot, labels=lab, buckets=bk)
    agree = (own.reshape(-1) == bk[1].reshape(-1)).float().mean().item()
    err = (lg - r_lg).abs()
    print(f'Reformer base, 2 hash rounds, T=4096: bucket agreement {agree:.4f}; |dlogit| max {err.max().item():.4f} '
          f'p99.9 {_quant(err, 0.999):.4f} mean {err.mean().item():.5f}; loss {o.loss.item():.5f} vs {r_loss.item():.5f}')
    assert agree > 0.97
    assert err.max().item() < 3e-2 and err.mean().item() < 5e-3
    assert abs(o.loss.item() - r_loss.item()) / r_loss.item() < 1e-3
