"""Round 6: the four-wave NT kernel (gemm_nt256w4_kernel) against the eight-wave one (MXL_GEMM_W4=0) at the C3 forward / dX shapes,
131072 tokens; TF/s.  NOCHECK=1 skips the comparison with torch.matmul (ablation builds compute garbage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT = int(os.environ.get('NT', 131072))
SHAPES = [('qkv fwd', 2304, 768), ('o', 768, 768), ('ffn1 fwd', 3072, 768), ('ffn2 fwd', 768, 3072), ('qkv dX', 768, 2304), ('K=8192', 2048, 8192)]
out = []
for name, N, K in SHAPES:
    torch.manual_seed(0)
    X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(2)]
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Y = [torch.empty(NT, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    for i in range(2): ops.gemm(X[i], W, Y[i], NT, N, K)
    torch.cuda.synchronize()
    if not os.environ.get('NOCHECK'):
        ref = torch.matmul(X[0][:4096].float(), W.float().t())
        err = (Y[0][:4096].float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-2, (name, err)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    s.record()
    for i in range(n): ops.gemm(X[i % 2], W, Y[i % 2], NT, N, K)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / n
    out.append(f'{name}:{2.0 * NT * N * K / t / 1e9:.0f}')
print(os.path.basename(os.environ.get('MXL_LIB_PATH', '') or 'default'), 'W4=' + os.environ.get('MXL_GEMM_W4', '1'), ' '.join(out), flush=True)
