"""Round 6: the weight-gradient GEMM dW += dY^T X (gemm_tt256_kernel) at the C3 and C4 layer shapes, 131072 tokens; us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd.xl_engine import XLEngine
dev = torch.device('cuda:0')
NT = 131072
out = []
for name, O, K in (('c3 qkv', 2304, 768), ('c3 o', 768, 768), ('c3 ffn1', 3072, 768), ('c3 ffn2', 768, 3072),
                   ('c4 d-d', 512, 512), ('c4 2d-d', 1024, 512), ('c4 F-d', 2048, 512), ('c4 d-F', 512, 2048)):
    X = torch.randn(NT, K, device=dev).bfloat16(); Y = (torch.randn(NT, O, device=dev) * 0.1).bfloat16()
    dW = torch.zeros(O, K, device=dev)
    ks = XLEngine._ks(O, K, NT)
    run = lambda: ops.gemm(Y, X, dW, O, K, NT, trans_a=True, trans_b=True, flags=ops.GEMM_OUT_F32_ATOMIC, ksplits=ks)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    out.append(f'{name}:{s.elapsed_time(e) / 20 * 1e3:.0f}')
print(os.path.basename(os.environ.get('MXL_LIB_PATH', '') or 'default'), ' '.join(out), flush=True)
