#!/usr/bin/env python3
"""Headline benchmark: Transformer-XL training throughput on synthetic token streams (BASELINE.json metric
"train tokens/sec/GPU (TransfoXL seq2048 bf16)"), one process per GPU, data-parallel over RCCL.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + backward + gradient all-reduce + clip + AdamW on one batch of (B, T) synthetic ids per GPU
(weak scaling).  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline     -- the dominant kernel, timed live with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (oracle/transfoxl_ref.py, the reference-style dense fp32 path) on the host cores,
                  rank 0 at N=1 only, on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[2] / SURVEY C3: the config the metric is quoted on (seq 2048); fits one GPU
    'c3': dict(name='TransfoXL 12L/768d H12 dh64 F3072 T=2048 M=2048 V=1190 cutoffs=[] (SURVEY C3, mode R: fresh zero mems)',
               size='base', n_layer=12, T=2048, M=2048, B=32),
    # BASELINE.json configs[1] / SURVEY C2
    'c2': dict(name='TransfoXL 6L/512d H8 dh64 F2048 T=1024 M=1024 V=1190 cutoffs=[] (SURVEY C2, mode R)',
               size='small', n_layer=6, T=1024, M=1024, B=64),
    'tiny': dict(name='debug 2L/128d T=256 M=256 (SURVEY C1 shape)', size='debug', n_layer=2, T=256, M=256, B=8),
}
V = 1190
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flops_per_token_fwd(L, d, T, M, Vv):
    """SURVEY 8(d), mode R: each query has M visible slots (BD needs all of them: 2dM), AC and PV (2d each per key) only the
    n_bar = (T+1)/2 real ones (T <= M); GEMMs 24 d^2; head 2 d V."""
    nbar = (T + 1) / 2 if T <= M else M
    return L * (24 * d * d + 4 * d * nbar + 2 * d * M) + 2 * d * Vv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3', choices=list(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (sequences)')
    ap.add_argument('--mode', default='train', choices=['train', 'decode', 'reformer'],
                    help='train = headline metric; decode = AR decode tok/s (SURVEY C5), single GPU replicas')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or os.environ.get('MXL_DIST_FORCE') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend='nccl', device_id=dev)
    else:
        dist = None

    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd.dist import GradSync
    from symbolic_music_generation_amd import ops

    if args.mode == 'decode':
        return decode_bench(args, dev, rank, world)
    if args.mode == 'reformer':
        return reformer_bench(args, dev, rank, world, dist)
    wl = WORKLOADS[args.workload]
    B = args.batch or wl['B']
    T, M = wl['T'], wl['M']
    cfg = MyTransfoXLConfig(wl['size'], max_length=T, vocab_size=V, n_layer=wl['n_layer'], mem_len=M, cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    eng = model.engine
    sync = GradSync(eng)
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)          # musicnlp/util/config.json "random-seed": 77
    ids = torch.randint(4, V, (B, T), generator=gen).to(dev)            # skip the special ids
    labels = ids.clone()
    # cosine schedule with warm-up, as TrainArgs defaults (train.py:165-190) -- lr value is irrelevant to throughput
    lr, wd = 3e-4, 0.1

    # ---- live roofline timing of the dominant kernel (relattn_bwd dkv+dq launches)
    timed = {'on': False, 'ev': []}
    orig_bwd = ops.relattn_bwd

    def timed_relattn_bwd(*a, **k):
        if timed['on']:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            finish = orig_bwd(*a, defer_drd=True, **k)          # the three attention-backward launches only
            e.record()
            timed['ev'].append((s, e))
            finish()                   # the dRd contraction (+ the r_r_bias gradient), outside the bracket
        else:
            orig_bwd(*a, **k)

    if not args.no_roofline:
        ops.relattn_bwd = timed_relattn_bwd

    def step():
        eng.zero_grad()
        model(input_ids=ids, labels=labels)
        eng.backward(layer_done=sync.layer_done)
        sync.finish()
        eng.optimizer_step(lr=lr, weight_decay=wd, max_grad_norm=1.0, grad_scale=1.0 / world)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timed['on'] = not args.no_roofline
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timed['on'] = False
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = model(input_ids=ids, labels=labels).loss.item()

    if rank == 0:
        tokens = B * T * world * args.steps
        d, L = cfg.d_model, cfg.n_layer
        f_fwd = flops_per_token_fwd(L, d, T, M, V)
        out = {
            'metric': 'train tokens/sec (TransfoXL seq2048 bf16), whole job', 'value': tokens / dt, 'unit': 'tokens/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': wl['name'], 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': T, 'mem_len': M,
                       'parallelism': f'dp{world}', 'dropout': cfg.dropout, 'final_loss': loss,
                       'train_flops_per_token': 3 * f_fwd,
                       'whole_step_mfma_frac': 3 * f_fwd * tokens / dt / world / (MFMA_BF16_PEAK_TFLOPS * 1e12)},
        }
        if not args.no_roofline and timed['ev']:
            ms = sum(s.elapsed_time(e) for s, e in timed['ev']) / len(timed['ev'])
            # algorithmic flops of ONE attention-backward call (one layer, this rank's batch): backward = 2 x forward of the
            # banded attention core, forward = B*T*(4*d*n_bar + 2*d*M)
            nbar = (T + 1) / 2 if T <= M else M
            alg = 2 * B * T * (4 * d * nbar + 2 * d * M)
            ach = alg / (ms * 1e-3) / 1e12
            out['roofline'] = {'kernel': 'relattn_bwd (delta + dq + dkv launches, one layer)', 'bound': 'mfma', 'achieved': ach,
                               'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / MFMA_BF16_PEAK_TFLOPS,
                               'traffic': pmc_traffic(args.workload, B), 'traffic_unit': 'HBM bytes per launch group (PMC)',
                               'avg_launch_ms': ms, 'launches_timed': len(timed['ev'])}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl, T, M)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def reformer_bench(args, dev, rank, world, dist):
    """Reformer 6L (3 local + 3 LSH) d=512 H=8 dh=64 F=2048, T=8192 (axial 64x128), num_hashes=1, per-GPU B=8,
    dropout on (0.05), auto num_buckets = [16,16].  Step = fwd + bwd + all-reduce + clip + AdamW."""
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    from symbolic_music_generation_amd.dist import GradSync
    B, T = args.batch or 16, 8192          # per-GPU batch sweep: 8 -> 3.61 M, 16 -> 3.87 M, 32 -> 3.97 M tok/s
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=T, axial_pos_shape=(64, 128), num_hashes=1)
    model = MyReformerModelWithLMHead(cfg, device=dev, seed=77).train()
    eng = model.engine
    sync = GradSync(eng)
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)
    ids = torch.randint(4, V, (B, T), generator=gen).to(dev)

    def step():
        eng.zero_grad()
        model(input_ids=ids, labels=ids)
        eng.backward(layer_done=sync.layer_done)
        sync.finish()
        eng.optimizer_step(lr=3e-4, weight_decay=0.01, max_grad_norm=1.0, grad_scale=1.0 / world)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        d = cfg.hidden_size
        # local layer 24d^2 + 2*2*(2*64)*d, LSH layer 22d^2 + n_h*(d*rot + 512 d), head 2*2d*V; train = 3x
        rot = 32
        f_fwd = 3 * (24 * d * d + 512 * d) + 3 * (22 * d * d + (d * rot + 512 * d)) + 2 * 2 * d * V
        tokens = B * T * world * args.steps
        print(json.dumps({'metric': 'train tokens/sec (Reformer 6L/512d seq8192 bf16), whole job', 'value': tokens / dt,
                          'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'bf16', 'data': 'synthetic',
                          'config': {'workload': 'SURVEY C4: Reformer small (3 local + 3 LSH) d=512 T=8192 axial 64x128 n_h=1',
                                     'per_gpu_batch': B, 'train_flops_per_token': 3 * f_fwd,
                                     'whole_step_mfma_frac': 3 * f_fwd * tokens / dt / world / (MFMA_BF16_PEAK_TFLOPS * 1e12)}}),
              flush=True)


def decode_bench(args, dev, rank, world):
    """TransfoXL 12L/768d cached-mem decode, batch 64 prompts x 256 tokens, top-k 8, generate to 2048,
    one hipGraph replay per token.  A 'step' here = one generated token for the whole batch."""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd.generate import XLDecoder
    B, Tp, M = args.batch or 64, 256, 2048
    cfg = MyTransfoXLConfig('base', max_length=2048, vocab_size=V, mem_len=M, cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).eval()
    dec = XLDecoder(model.engine, B, 2048, seed=77 + rank)
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)
    prompt = torch.randint(4, V, (B, Tp), generator=gen).to(dev)
    samp = dict(do_sample=True, top_k=8, top_p=1.0, temperature=1.0)
    dec.prefill(prompt, samp)
    for _ in range(max(args.warmup, 1)):
        dec.step(samp)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dec.step(samp)
    torch.cuda.synchronize()
    # default: the whole C5 generation, prompt 256 -> T = 2048 (the first steps are cheaper than the last: ring slots that were
    # never written are HF's zero mems and cost no K/V bytes, so a short window after the prompt would flatter the number)
    done = Tp + 1 + max(args.warmup, 1) + 1                 # positions filled so far: prompt, its sample, warm-up and capture steps
    steps = args.steps if args.steps != 10 else 2048 - done
    steps = min(steps, 2048 - done)
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d, L = cfg.d_model, cfg.n_layer
    # algorithmic bytes per step: weights once + the WRITTEN part of the projected K/V ring per sequence, averaged
    # over the timed steps (+ Rd tables, L2-resident); `full_ring` is the figure for a full memory (2 M d 2 B per layer)
    valid = sum(min(done + i, M) for i in range(steps)) / max(steps, 1)
    bytes_step = L * 12 * d * d * 2 + V * d * 2 + B * L * 2 * valid * d * 2
    bytes_full = L * 12 * d * d * 2 + V * d * 2 + B * L * 2 * M * d * 2
    if rank == 0:
        out = {'metric': 'AR decode tokens/sec (TransfoXL 12L/768d, batch 64, cached mems, top-k 8, hipGraph step)',
               'value': B * steps / dt, 'unit': 'tokens/s', 'n_gpus': 1, 'steps': steps, 'warmup': args.warmup,
               'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': 'SURVEY C5 decode: 12L/768d, M=2048, B=64 prompts x 256 tokens generated to T=2048, top_k=8',
                          'batch': B, 'positions_timed': [done, done + steps]},
               'roofline': {'kernel': 'whole decode step (hipGraph replay)', 'bound': 'hbm',
                            'achieved': bytes_step / (dt / steps) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                            'frac': bytes_step / (dt / steps) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                            'algorithmic_bytes_per_step': bytes_step, 'algorithmic_bytes_per_step_full_ring': bytes_full,
                            'mean_valid_ring_slots': valid}}
        print(json.dumps(out), flush=True)


def pmc_traffic(workload, B):
    """HBM bytes per attention-backward call from the committed PMC passes (profiles/r01_c3_pmc_traffic.json, produced by
    scripts/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE rocprofv3 runs of this same command; FETCH doubled per the
    gfx950 correction).  Counters cannot be collected from inside the timed run, so this is the recorded measurement for the
    default workload and None for any other."""
    path = os.path.join(ROOT, 'profiles', 'r01_c3_pmc_traffic.json')
    if workload != 'c3' or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get('per_gpu_batch', 16) != B:          # the passes were collected at one batch size; no figure for another
        return None
    k = rec['kernels']
    dq = 'relattn_bwd_dq8_kernel<64>' if 'relattn_bwd_dq8_kernel<64>' in k else 'relattn_bwd_dq_kernel<64>'
    names = ('relattn_bwd_delta_kernel', dq, 'relattn_bwd_dkv_kernel<64>')
    if not all(n in k for n in names):
        return None
    return sum(k[n]['hbm_bytes_per_launch'] for n in names)


def cpu_baseline(wl, T, M):
    """The oracle (reference-style dense fp32 TransfoXL, oracle/transfoxl_ref.py) on the host cores: train steps
    (fwd + bwd + clip + AdamW) on a bounded sample: B=1 sequence of the same T/M and, to stay within ~30 s of CPU work,
    at most 3 of the workload's layers (per-layer cost is identical; embedding + head are included once), scaled to the
    full depth in `value`."""
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    torch.manual_seed(77)
    L_full = wl['n_layer']
    L_s = min(L_full, 3)
    c = RefXLConfig.from_preset(wl['size'], vocab_size=V, max_length=T, mem_len=M, cutoffs=[], n_layer=L_s)
    m = RefTransfoXLLMHeadModel(c).train()
    ids = torch.randint(4, V, (1, T))
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=0.1)
    times = []
    for i in range(2):
        t = time.perf_counter()
        o = m(ids, labels=ids)
        o.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t)
    t_full = times[-1] * L_full / L_s
    return {'value': T / t_full, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 train step after 1 warm-up, B=1 x T={T} (M={M}), fp32, {L_s} of {L_full} layers timed '
                      f'({times[-1]:.1f} s) and scaled x{L_full}/{L_s}'}


if __name__ == '__main__':
    main()
