"""Round 6: mxl_axial_embed_bwd at the C4 shape (16 x 8192 tokens, V 1190, d 512), LDS-accumulating form against the one-atomic-per-element
form (MXL_AXIAL_BWD_GLOBAL=1), with the training step's dropout and second gradient stream; us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
dev = torch.device('cuda:0')
B, T, V, d, A0, A1, d0 = 16, 8192, 1190, 512, 64, 128, 128
torch.manual_seed(0)
ids = torch.randint(4, V, (B, T), device=dev)
dout = torch.randn(B, T, d, device=dev).bfloat16(); dout2 = torch.randn(B, T, d, device=dev).bfloat16()
dE = torch.zeros(V, d, device=dev); dW0 = torch.zeros(A0, d0, device=dev); dW1 = torch.zeros(A1, d - d0, device=dev)
for form in ('lds', 'global', 'lds', 'global'):
    if form == 'global': os.environ['MXL_AXIAL_BWD_GLOBAL'] = '1'
    else: os.environ.pop('MXL_AXIAL_BWD_GLOBAL', None)
    run = lambda: ops.axial_embed_bwd(ids, dout, dE, dW0, dW1, A0, A1, drop_p=0.05, seed=3, site_emb=1, site_pos=2, dout2=dout2)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print(f'{form:7s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us', flush=True)
