"""`mxl_set_reserved_cus(k)` (include/musicxl.h; `dist.GradSync` sets it from MXL_RESERVE_CUS when a gradient exchange is live,
SURVEY 8(e)): the persistent GEMM grids then run on n_cu - k workgroups, which changes the NT kernel's grid, its 256 / 192 tile
width choice and the weight-gradient kernel's split-K factor.  None of that may change results: NT outputs are bit-identical
(every output element is one workgroup's K loop in one order, whatever the grid), split-K weight gradients differ by fp32
summation order only.  Checked at the persistent kernels' own shapes and through one training step of the C3 layer shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

K_RESERVED = 16


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.fixture
def reserve(dev):
    from symbolic_music_generation_amd import ops

    def set_k(k):
        torch.cuda.synchronize()
        ops.set_reserved_cus(k)
    yield set_k
    torch.cuda.synchronize()
    ops.set_reserved_cus(0)


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (1000, 300, 192), (2048 + 17, 2304, 768), (4096, 1190, 768), (16384, 768, 3072),
                                   (16384, 3072, 768)])
def test_nt_gemm_bit_exact_under_reserved_cus(dev, reserve, M, N, K):
    """the persistent 256 x {256, 192} NT kernel, plain and with every compile-time epilogue of the engines, k = 0 against k = 16"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(M + N + K)
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)

    def run():
        c32 = torch.empty(M, N, device=dev, dtype=torch.float32)
        ops.gemm(x, w, c32, M, N, K, flags=ops.GEMM_OUT_F32)
        c16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(x, w, c16, M, N, K)
        y = ops.linear(x, w, b, relu=True)
        y2 = ops.linear(x, w, b, relu=True, drop_p=0.25, seed=5, site=3)
        return c32, c16, y, y2

    reserve(0)
    base = run()
    reserve(K_RESERVED)
    got = run()
    for a, g in zip(base, got):
        assert torch.equal(a, g)
    assert _rel(base[0], x.float() @ w.float().t()) < 1e-5


@pytest.mark.parametrize('O,K,NTOK', [(768, 768, 32768), (3072, 768, 32768), (768, 3072, 16384), (2304, 768, 16384)])
def test_weight_gradient_split_k_under_reserved_cus(dev, reserve, O, K, NTOK):
    """dW = dY^T X on the persistent transposed-read kernel: the split-K factor follows the free CU count, the sum is the same to
    fp32 reordering, and the column / row checksums (fp64, from the same bf16 operands) hold under both"""
    from symbolic_music_generation_amd import ops
    torch.manual_seed(O + K)
    x = (torch.randn(NTOK, K, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(NTOK, O, device=dev) * 0.5).to(torch.bfloat16)

    def run():
        dw = torch.zeros(O, K, device=dev, dtype=torch.float32)
        ops.gemm(dy, x, dw, O, K, NTOK, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=4)
        return dw

    reserve(0)
    a = run()
    reserve(K_RESERVED)
    b = run()
    assert _rel(b, a) < 2e-6
    col = dy.double().sum(1) @ x.double()
    assert _rel(a.double().sum(0), col) < 1e-5 and _rel(b.double().sum(0), col) < 1e-5


def test_c3_shape_train_step_under_reserved_cus(dev, reserve):
    """one forward + backward of the C3 layer shapes (768d / H12 / dh64 / F3072, T = M = 2048, two layers, batch 4, dropout on)
    with k = 16 against k = 0 from the same parameters and rng_step: the forward is NT GEMMs and attention only, so the loss agrees to
    the last bits of its own atomic reduction; gradients agree to the atomics' reordering (the bound test_c3_bench_batch_dropout_step_... uses for two runs
    of the SAME configuration)"""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    T = M = 2048
    V = 1190
    cfg = MyTransfoXLConfig('base', max_length=T, vocab_size=V, n_layer=2, mem_len=M, cutoffs=[])
    m = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    eng = m.engine
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(4, V, (4, T), generator=g).to(dev)

    def step():
        eng.rng_step = 3
        with torch.no_grad():
            eng.zero_grad()
            o = m(input_ids=ids, labels=ids)
            eng.backward()
        torch.cuda.synchronize()
        return o.loss.item(), eng.G.clone()

    def worst(ga, gb):
        w, wn = 0.0, ''
        for name in eng.layout.real_names():
            a, b = eng.layout.view(ga, name).double(), eng.layout.view(gb, name).double()
            e = ((a - b).norm() / (a.norm() + 1e-30)).item()
            if e > w:
                w, wn = e, name
        return w, wn

    reserve(0)
    l0, g0 = step()
    l0b, g0b = step()                           # the same configuration again: what the float atomics alone move
    reserve(K_RESERVED)
    l1, g1 = step()
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)      # (the loss reduction itself sums with float atomics)
    noise, _ = worst(g0, g0b)
    w, wn = worst(g0, g1)
    print(f'k = 0 twice: {noise:.2e}; k = {K_RESERVED} against k = 0: {w:.2e} ({wn})')
    assert w < max(3.0 * noise, 3e-4), (w, wn, noise)
