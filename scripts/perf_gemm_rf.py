"""NT GEMM at the Reformer (C4) layer shapes, d = 512, F = 2048, 131072 tokens: this library vs torch.matmul (hipBLASLt, a cross-check
only) on the same operands, with the HBM floor of each shape (operands + result once at 6 TB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
NT = int(os.environ.get('NT', 131072))

def timeit(fn, n=20, warm=3):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

NB = 4
for name, O, K in (('d->d', 512, 512), ('d->2d', 1024, 512), ('d->3d', 1536, 512), ('ffn1 d->F', 2048, 512), ('ffn2 F->d', 512, 2048),
                   ('C2 d->d (768 tok x)', 512, 512)):
    X = [torch.randn(NT, K, device=dev).bfloat16() for _ in range(NB)]
    W = torch.randn(O, K, device=dev).bfloat16() * 0.05
    Y = [torch.empty(NT, O, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
    fl = 2.0 * NT * O * K
    floor_us = (NT * K + NT * O + O * K) * 2 / 6e12 * 1e6
    t = timeit(lambda i: ops.gemm(X[i % NB], W, Y[i % NB], NT, O, K))
    tt = timeit(lambda i: torch.matmul(X[i % NB], W.t(), out=Y[i % NB]))
    print(f'{name:20s} [{NT}x{O}x{K}]: ours {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch {tt*1e3:7.1f} us {fl/tt/1e9:6.0f} TF/s | HBM floor {floor_us:6.1f} us, MFMA floor {fl/2.5e15*1e6:6.1f} us', flush=True)
