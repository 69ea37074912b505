"""Weight-gradient GEMM dW = dY^T X at the C3 shapes (131072 tokens): time + a correctness check against fp32 torch.matmul on the
same bf16 operands.  MXL_GEMM_NO_TT256=1 selects the 128 x 128 split-K kernel for an A/B in a second process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd.xl_engine import XLEngine
dev = torch.device('cuda:0')
NT, d, F = int(os.environ.get('NT', 131072)), 768, 3072
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.manual_seed(0)
for name, O, K in (('qkv', 3 * d, d), ('o', d, d), ('ffn1', F, d), ('ffn2', d, F)):
    X = torch.randn(NT, K, device=dev).bfloat16(); Y = (torch.randn(NT, O, device=dev) * 0.1).bfloat16()
    dW = torch.zeros(O, K, device=dev)
    ks = XLEngine._ks(O, K, NT)
    run = lambda: ops.gemm(Y, X, dW, O, K, NT, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=ks)
    run(); torch.cuda.synchronize()
    ref = Y[:, :256].float().t() @ X.float()            # first 256 rows of dW
    ref2 = Y.float()[:, -64:].t() @ X.float()
    err = max(((dW[:256] - ref).norm() / ref.norm()).item(), ((dW[-64:] - ref2).norm() / ref2.norm()).item())
    dW.zero_()
    t = timeit(run)
    print(f'{name:5s} dW [{O}x{K}x{NT}] ks={ks}: {t*1e3:7.1f} us {2.0*NT*O*K/t/1e9:6.0f} TF/s   rel err vs fp32 {err:.2e}', flush=True)
