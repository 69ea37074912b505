"""Per-shape GEMM table at the C3 layer shapes: this library vs torch.matmul (hipBLASLt) on the same operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT = int(os.environ.get('NT', 131072))          # tokens: per-GPU batch 64 x 2048 (round 1 ran this at 32768)
d, F = 768, 3072

def timeit(fn, n=20, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def ks(m, n):
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    return max(1, min(8, 512 // max(tiles, 1)))

NB = 4
rows = []
for name, O, K in (('qkv', 3 * d, d), ('o', d, d), ('ffn1', F, d), ('ffn2', d, F)):
    X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(NB)]
    W = torch.randn(O, K, device=dev).bfloat16() * 0.05
    Y = [torch.empty(NT, O, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    dX = [torch.empty(NT, K, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    dW = torch.zeros(O, K, device=dev, dtype=torch.float32)
    fl = 2.0 * NT * O * K
    t = timeit(lambda i: ops.gemm(X[i % NB], W, Y[i % NB], NT, O, K))
    tt = timeit(lambda i: torch.matmul(X[i % NB], W.t(), out=Y[i % NB]))
    print(f'{name:5s} fwd  Y=X W^T   [{NT}x{O}x{K}]: ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch {tt*1e3:7.1f} us {fl/tt/1e9:6.0f} TF/s', flush=True)
    Wt = W.t().contiguous()       # the engines keep [in][out] copies of the weights: the input gradient is an NT product too
    t = timeit(lambda i: ops.gemm(Y[i % NB], Wt, dX[i % NB], NT, K, O))
    tt = timeit(lambda i: torch.matmul(Y[i % NB], W, out=dX[i % NB]))
    if os.environ.get('SKIP_DW'):
        print(f'{name:5s} dX   dX=dY W   [{NT}x{K}x{O}]: ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s', flush=True); continue
    print(f'{name:5s} dX   dX=dY W   [{NT}x{K}x{O}]: ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch {tt*1e3:7.1f} us {fl/tt/1e9:6.0f} TF/s', flush=True)
    k_ = ks(O, K)
    t = timeit(lambda i: ops.gemm(Y[i % NB], X[i % NB], dW, O, K, NT, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=k_))
    dWb = torch.empty(O, K, device=dev, dtype=torch.bfloat16)
    tt = timeit(lambda i: torch.matmul(Y[i % NB].t(), X[i % NB], out=dWb))
    print(f'{name:5s} dW   dW=dY^T X [{O}x{K}x{NT}] ks={k_}: ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch {tt*1e3:7.1f} us {fl/tt/1e9:6.0f} TF/s', flush=True)
