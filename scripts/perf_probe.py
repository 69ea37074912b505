"""Ad-hoc kernel timing on the GPU box (not part of the test suite)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops

dev = torch.device('cuda:0')

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def gemm_case(M, N, K, ta=False, tb=False, f32=False, ks=1):
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((K, N) if tb else (N, K), device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    flags = ops.GEMM_OUT_F32_ATOMIC if f32 else 0
    ms = timeit(lambda: ops.gemm(a, b, c, M, N, K, trans_a=ta, trans_b=tb, flags=flags, ksplits=ks))
    print(f'gemm M={M} N={N} K={K} ta={ta} tb={tb} ks={ks}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s', flush=True)

N = 32768
for (n, k) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    gemm_case(N, n, k)
    gemm_case(N, k, n, tb=True)
    gemm_case(n, k, N, ta=True, tb=True, f32=True, ks=8)
gemm_case(8192, 8192, 8192)

def attn_case(B, T, H, dh, M, Kc):
    d = H * dh
    qkv = torch.randn(B, Kc, 3 * d, device=dev).bfloat16()
    rd = torch.randn(M, d, device=dev).bfloat16()
    rwb = torch.randn(H, dh, device=dev) * .1; rrb = torch.randn(H, dh, device=dev) * .1
    out = torch.zeros(B, T, d, device=dev, dtype=torch.bfloat16); lse = torch.zeros(B, H, T, device=dev)
    f = lambda: ops.relattn_fwd(qkv[:, Kc - T:, :d], qkv[:, :, d:2*d], qkv[:, :, 2*d:], rd, rwb, rrb, out, lse, B=B, T=T, H=H, dh=dh,
                                M=M, Kc=Kc, q_bs=Kc*3*d, q_rs=3*d, kv_bs=Kc*3*d, kv_rs=3*d, rd_rs=d, o_bs=T*d, o_rs=d)
    ms = timeit(f, n=10)
    # algorithmic flops: per token 2*d*M (BD) + 4*d*nbar (AC+PV), nbar = real keys per query
    nbar = M if Kc == M + T else (T + 1) / 2 if T <= M else None
    fl = B * T * (2 * d * M + 4 * d * nbar)
    print(f'relattn_fwd B={B} T={T} H={H} dh={dh} M={M} Kc={Kc}: {ms:.3f} ms  alg {fl/ms/1e9:.1f} TFLOP/s', flush=True)

attn_case(16, 2048, 12, 64, 2048, 2048)
attn_case(16, 2048, 12, 64, 2048, 4096)
attn_case(32, 1024, 8, 64, 1024, 1024)
