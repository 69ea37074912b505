"""decode_attn_kernel alone (one layer, full ring or `NV` valid slots): microseconds and streamed TB/s, B = 16 / 32 / 64 rows,
ring pieces per (sequence, head) 1 / 2 / 4 (mxl_relattn_decode_split) and the largest difference of the outputs from the one-piece form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from symbolic_music_generation_amd import ops
from symbolic_music_generation_amd.ops import lib, _p, _stream, check
dev = torch.device('cuda:0')
H, dh, M = 12, 64, 2048
d = H * dh
for B in [int(x) for x in os.environ.get("BS", "16,32,64").split(",")]:
    for nv in [int(x) for x in os.environ.get('NVS', f"{os.environ.get('NV', M)},1153").split(',')]:
        NB = 6                                   # rotate ring buffers so that nothing is served from the Infinity Cache
        kc = [torch.randn(B, H, M, dh, device=dev).bfloat16() for _ in range(NB)]
        vc = [torch.randn(B, H, M, dh, device=dev).bfloat16() for _ in range(NB)]
        qkv = torch.randn(B, 3 * d, device=dev).bfloat16()
        bd = torch.randn(B, H, M, device=dev)
        rwb = torch.randn(H, dh, device=dev) * .1
        out = torch.empty(B, d, device=dev, dtype=torch.bfloat16)
        t_dev = torch.tensor([nv - 1 if nv < M else 3 * M + 5], device=dev, dtype=torch.int32)
        ref = None
        for pieces in [int(x) for x in os.environ.get('PIECES', '1,2,4').split(',')]:
            sp = ops.relattn_decode_split_scratch(B, H, dh, pieces, dev)
            def run(i):
                if pieces == 1:
                    check(lib().mxl_relattn_decode(_p(qkv), _p(kc[i % NB]), _p(vc[i % NB]), _p(bd), _p(rwb), _p(out), _p(t_dev), B, H, dh, M,
                                                   0.125, _stream()), 'decode')
                else:
                    check(lib().mxl_relattn_decode_split(_p(qkv), _p(kc[i % NB]), _p(vc[i % NB]), _p(bd), _p(rwb), _p(out), _p(t_dev), B, H,
                                                         dh, M, 0.125, pieces, _p(sp[0]), _p(sp[1]), _stream()), 'decode split')
            run(0); torch.cuda.synchronize()
            o0 = out.float().clone()
            if ref is None: ref = o0
            for i in range(3): run(i)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            s.record()
            for i in range(n): run(i)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / n * 1e3
            byt = B * H * min(nv, M) * dh * 2 * 2
            print(f'B={B:3d} valid slots {min(nv, M):5d} pieces {pieces}: {us:7.1f} us  {byt / us / 1e6:5.2f} TB/s   max |out - one piece| '
                  f'{(o0 - ref).abs().max().item():.2e}', flush=True)
