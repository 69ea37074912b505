"""A/B of the persistent NT GEMM: run once per library build (MXL_LIB_PATH selects it), interleave the runs from the shell.
Prints TF/s per C3 shape; with --screen also repeats each shape and checks run-to-run bit equality and the result against
torch.matmul (race screen for schedule edits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT = int(os.environ.get('NT', 32768))
SHAPES = [('qkv fwd', 2304, 768), ('o fwd/dX', 768, 768), ('ffn1 fwd', 3072, 768), ('ffn2 fwd', 768, 3072), ('qkv dX', 768, 2304),
          ('head-ish', 1216, 768), ('K=8192', 2048, 8192)]
screen = '--screen' in sys.argv
tag = os.environ.get('MXL_LIB_PATH', 'default')
out = []
for name, N, K in SHAPES:
    torch.manual_seed(0)
    X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(3)]
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Y = [torch.empty(NT, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    for i in range(3): ops.gemm(X[i], W, Y[i], NT, N, K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    s.record()
    for i in range(n): ops.gemm(X[i % 3], W, Y[i % 3], NT, N, K)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / n
    out.append(f'{name}:{2.0 * NT * N * K / t / 1e9:.0f}')
    if screen:
        ref = torch.matmul(X[0].float(), W.float().t())
        first = None
        for r in range(40):
            Y[0].zero_()
            ops.gemm(X[0], W, Y[0], NT, N, K)
            torch.cuda.synchronize()
            if first is None:
                first = Y[0].clone()
                err = (first.float() - ref).abs().max().item() / ref.abs().max().item()
                assert err < 1e-2, (name, err)
            else:
                assert torch.equal(first, Y[0]), (name, r)
print(os.path.basename(tag), ' '.join(out), 'screen ok' if screen else '', flush=True)
