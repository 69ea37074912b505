"""Which hipBLASLt kernels torch.matmul picks at the C3 linear shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
dev = torch.device('cuda:0')
NT = 32768
for name, N, K in [('qkv', 2304, 768), ('o', 768, 768), ('ffn1', 3072, 768), ('ffn2', 768, 3072), ('qkvdx', 768, 2304)]:
    X = torch.randn(NT, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Y = torch.empty(NT, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(X, W.t(), out=Y)
    torch.cuda.synchronize()
