"""Round 6: the small launches of one C5 decode layer (12L / 768d, lane of B rows) timed one by one -- each op replayed back to back
inside ONE hipGraph of `REP` nodes (stream order: a launch starts when the previous one has ended, so time / REP = duration + the
graph's node-to-node gap, what the step's dependent chain pays).  B=32 python scripts/perf_decode_ops.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symbolic_music_generation_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(os.environ.get('B', '32'))
REP = int(os.environ.get('REP', '200'))
d, Fi, H, dh, M = 768, 3072, 12, 64, 2048
bf = dict(device=dev, dtype=torch.bfloat16)
torch.manual_seed(0)
x = torch.randn(B, d, **bf); res = torch.randn(B, d, **bf); a = torch.randn(B, Fi, **bf)
# many copies of the weights (the step streams 170 MB of them: they are not L2-resident when their launch starts)
NW = 24
wqkv = [torch.randn(3 * d, d, **bf) * 0.05 for _ in range(NW)]
wo = [torch.randn(d, d, **bf) * 0.05 for _ in range(NW)]
w1 = [torch.randn(Fi, d, **bf) * 0.05 for _ in range(NW)]
w2 = [torch.randn(d, Fi, **bf) * 0.05 for _ in range(NW)]
b1 = torch.randn(Fi, device=dev); b2 = torch.randn(d, device=dev)
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
qkv = torch.empty(B, 3 * d, **bf); out_d = torch.empty(B, d, **bf); out_f = torch.empty(B, Fi, **bf); y = torch.empty(B, d, **bf)
slabs = torch.zeros(4, 64, d, device=dev, dtype=torch.float32)
kc = torch.zeros(B, H, M, dh, **bf); vc = torch.zeros(B, H, M, dh, **bf)
t_dev = torch.tensor([1153], device=dev, dtype=torch.int32)
rrb = torch.randn(d, device=dev); qr = torch.empty(B, d, **bf)
rd = torch.randn(M, d, **bf); bd = torch.empty(B, H, M, device=dev, dtype=torch.float32)


def timeit(name, fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REP):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    print(f'{name:58s} {best:7.2f} us', flush=True)


G = ops.gemm_skinny
timeit('qkv + append (N 2304, K 768)', lambda i: ops.decode_qkv(x, wqkv[i % NW], qkv, kc, vc, t_dev, rrb, qr, dh))
timeit('positional term bd', lambda i: ops.check(ops.lib().mxl_decode_bd(ops._p(qr), ops._p(rd), ops._p(bd), B, H, dh, M, d, d, ops._stream()), 'bd'))
timeit('o projection (N 768, K 768)', lambda i: G(x, wo[i % NW], out_d, B, d, d))
timeit('LayerNorm(x + res)', lambda i: ops.ln_residual_fwd(x, res, gam, bet, y, eps=1e-5))
timeit('ffn1 (N 3072, K 768) bias relu', lambda i: G(x, w1[i % NW], out_f, B, Fi, d, flags=ops.GEMM_BIAS | ops.GEMM_RELU, bias=b1))
timeit('ffn2 (N 768, K 3072) four K-slices -> slabs', lambda i: ops.gemm_skinny_partial(a, w2[i % NW], slabs, B, d, Fi, 4))
timeit('LayerNorm(res + slabs + bias)', lambda i: ops.ln_residual_fwd_partial(slabs, 4, b2, res, gam, bet, y, eps=1e-5))
timeit('ffn2 unsliced, bias', lambda i: G(a, w2[i % NW], out_d, B, d, Fi, flags=ops.GEMM_BIAS, bias=b2))
