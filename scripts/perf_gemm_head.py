"""The heads' logits GEMM (fp32 output + bias, ragged N) against the same product with bf16 output and against hipBLASLt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=10, warm=2):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for M, V, K in ((65536, 420, 1024), (131072, 1190, 768), (65536, 512, 1024)):
    ld = (V + 7) // 8 * 8
    X = torch.randn(M, K, device=dev).bfloat16()
    W = torch.randn(ld, K, device=dev).bfloat16() * 0.05
    b = torch.randn(V, device=dev)
    Yf = torch.empty(M, ld, device=dev, dtype=torch.float32)
    Yb = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.gemm(X, W, Yf, M, V, K, flags=ops.GEMM_OUT_F32 | ops.GEMM_BIAS, bias=b))
    t2 = timeit(lambda: ops.gemm(X, W, Yf, M, V, K, flags=ops.GEMM_OUT_F32))
    t3 = timeit(lambda: ops.gemm(X, W, Yb, M, V, K, flags=ops.GEMM_BIAS, bias=b))
    t4 = timeit(lambda: ops.gemm(X, W, Yb, M, V, K))
    t5 = timeit(lambda: torch.matmul(X, W[:V].t(), out=Yb[:, :V]) if ld == V else torch.matmul(X, W.t(), out=Yb))
    print(f'[{M} x {V} x {K}] f32+bias {t1*1e3:7.1f} us  f32 {t2*1e3:7.1f}  bf16+bias {t3*1e3:7.1f}  bf16 {t4*1e3:7.1f}  hipBLASLt bf16 {t5*1e3:7.1f}', flush=True)
