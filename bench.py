#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json metric "train tokens/sec/GPU (TransfoXL seq2048 bf16) at 1/2/4/8 GPUs; AR decode tok/s").

    python bench.py --gpus N --steps K --warmup W

* N > 1 and no launcher environment: this process touches no GPU; it starts N ranks with
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one process per GPU, RCCL),
  relays rank 0's JSON line and exits with the children's code.  Started under a launcher (WORLD_SIZE set) it is one rank.
* The headline leg (`value`) is the SURVEY C3 training step: forward + backward + gradient all-reduce + clip + AdamW on one
  batch of (B, T) synthetic ids per GPU (weak scaling).  At N = 1 the same process then runs the other two legs of the
  metric and attaches them to the same JSON line: `"decode"` (cached-mem AR decode, batch 64, top-k 8, one
  hipGraph replay per token) and `"reformer"` (Reformer 6L/512d, T = 8192).  Decode and eval are replicas only
  (DESIGN 6), so at N > 1 only the training leg runs.
* Every leg carries `roofline` (dominant kernel, HIP events on the launch stream inside the timed region) and, at N = 1,
  `cpu_baseline`: the CPU oracle (oracle/*.py, the reference-style fp32 path; `kind: "port"`) on the host cores on a
  bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402   (importing torch does not initialise the GPU)

WORKLOADS = {
    # BASELINE.json configs[2] / SURVEY C3: the config the metric is quoted on (seq 2048); fits one GPU
    'c3': dict(name='TransfoXL 12L/768d H12 dh64 F3072 T=2048 M=2048 V=1190 cutoffs=[] (SURVEY C3, mode R: fresh zero mems)',
               size='base', n_layer=12, T=2048, M=2048, B=64),
    # BASELINE.json configs[1] / SURVEY C2
    'c2': dict(name='TransfoXL 6L/512d H8 dh64 F2048 T=1024 M=1024 V=1190 cutoffs=[] (SURVEY C2, mode R)',
               size='small', n_layer=6, T=1024, M=1024, B=64),
    'tiny': dict(name='debug 2L/128d T=256 M=256 (SURVEY C1 shape)', size='debug', n_layer=2, T=256, M=256, B=8),
}
V = 1190
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
CPU_BUDGET_S = 30.0              # bounded CPU-oracle sample per leg


def flops_per_token_fwd(L, d, T, M, Vv):
    """SURVEY 8(d), mode R: each query has M visible slots (BD needs all of them: 2dM), AC and PV (2d each per key) only the
    n_bar = (T+1)/2 real ones (T <= M); GEMMs 24 d^2; head 2 d V."""
    nbar = (T + 1) / 2 if T <= M else M
    return L * (24 * d * d + 4 * d * nbar + 2 * d * M) + 2 * d * Vv


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3', choices=list(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (sequences)')
    ap.add_argument('--mode', default='all', choices=['all', 'train', 'decode', 'reformer'],
                    help='all = train leg + (at one GPU) the decode and Reformer legs on the same JSON line')
    ap.add_argument('--eager', action='store_true', help='decode leg without hipGraph replay (PMC passes: rocprofv3 cannot '
                                                         'collect counters over graph replays)')
    ap.add_argument('--decode-steps', type=int, default=0, help='decode steps to time (0 = the whole C5 generation)')
    ap.add_argument('--decode-prompt', type=int, default=256, help='prompt length of the decode leg (256 = SURVEY C5; the PMC passes use a\n'
                    '                    longer one so that their short eager window sits at the mean ring occupancy of the generation)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-context-legs', action='store_true', help='skip the C2 / mode-S / published-config legs of the default line')
    ap.add_argument('--grad-exchange', default='fp32', choices=['fp32', 'bf16'],
                    help='dtype of the data-parallel gradient all-reduce (fp32 = what HF DDP exchanges in the reference stack)')
    ap.add_argument('--master-port', type=int, default=29533)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_argv(args, argv):
    """The child command for `--gpus N` (N > 1): the launch line the driver itself uses."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(args.master_port), os.path.abspath(__file__)] + list(argv)


def spawn(args, argv) -> int:
    """Parent of an N-rank run: no GPU call happens in this process (a process that has initialised the GPU must not
    start or replace programs on this pool).  Children inherit stdout; only rank 0 prints the JSON line."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL on this pool
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.Popen(spawn_argv(args, argv), env=env)
    return proc.wait()


# ------------------------------------------------------------------------------------------------ shared timing skeleton
class Ranks:
    """process-group plumbing shared by the legs (and by the CPU stub used in tests)"""

    def __init__(self, backend, dev):
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL on this pool, also under an external torchrun
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.dev = dev
        self.dist = None
        if self.world > 1 or os.environ.get('MXL_DIST_FORCE') == '1':
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
            kw = dict(device_id=dev) if backend == 'nccl' else {}
            dist.init_process_group(backend=backend, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.dev.type == 'cuda':
            torch.cuda.synchronize()

    def max_over_ranks(self, dt: float) -> float:
        if self.dist is None:
            return dt
        t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def rccl_ranks(self) -> int:
        return self.dist.get_world_size() if self.dist is not None else 1

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_steps(ranks: Ranks, step, steps: int, warmup: int, on_timed=None) -> float:
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        step()
    ranks.barrier()
    if on_timed is not None:
        on_timed(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ranks.barrier()
    dt = time.perf_counter() - t0
    if on_timed is not None:
        on_timed(False)
    return ranks.max_over_ranks(dt)


class EventBracket:
    """HIP events on torch's current stream -- the stream every libmusicxl launch of this process goes to (ops._stream)."""

    def __init__(self):
        self.on, self.ev, self.work = False, [], 0.0

    def run(self, fn, work=0.0):
        if not self.on:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.ev.append((s, e))
        self.work += work
        return r

    def total_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.ev)


# ------------------------------------------------------------------------------------------------ legs
def train_leg(args, ranks: Ranks):
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd.dist import GradSync
    from symbolic_music_generation_amd import ops
    dev, rank, world = ranks.dev, ranks.rank, ranks.world
    wl = WORKLOADS[args.workload]
    B = args.batch or wl['B']
    T, M = wl['T'], wl['M']
    cfg = MyTransfoXLConfig(wl['size'], max_length=T, vocab_size=V, n_layer=wl['n_layer'], mem_len=M, cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    eng = model.engine
    sync = GradSync(eng, dtype=args.grad_exchange)
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)          # musicnlp/util/config.json "random-seed": 77
    ids = torch.randint(4, V, (B, T), generator=gen).to(dev)            # skip the special ids
    labels = ids.clone()
    lr, wd = 3e-4, 0.1            # train_xl's weight decay (train.py:570); the lr value is irrelevant to throughput

    # live roofline timing of the dominant kernel group: EVERY launch of one layer's attention backward (delta, query-owner,
    # key-owner, the q + r_r_bias operand, the dRd contraction) inside one HIP-event bracket on the launch stream, plus the
    # library's own per-kernel event pairs (mxl_ktime_*) for the forward and each backward kernel
    br = EventBracket()
    orig_bwd, orig_fused = ops.relattn_bwd, ops.relattn_bwd_fused

    def timed_relattn_bwd(*a, **k):
        return br.run(lambda: orig_bwd(*a, **k))

    def timed_relattn_bwd_fused(*a, **k):          # round 4: delta + fused pass + dq finish + (zero memories) the phantom cells' dRd
        return br.run(lambda: orig_fused(*a, **k))

    if not args.no_roofline:
        ops.relattn_bwd = timed_relattn_bwd
        ops.relattn_bwd_fused = timed_relattn_bwd_fused

    def on_timed(on):
        br.on = on and not args.no_roofline
        if not args.no_roofline:
            ops.ktime_enable(on)

    def step():
        with torch.no_grad():          # the fused path: explicit engine backward, no autograd node needed
            eng.zero_grad()
            model(input_ids=ids, labels=labels)
            eng.backward(layer_done=sync.layer_done)
            sync.finish()
            eng.optimizer_step(lr=lr, weight_decay=wd, max_grad_norm=1.0, grad_scale=1.0 / world)

    dt = timed_steps(ranks, step, args.steps, args.warmup, on_timed=on_timed)
    ops.relattn_bwd, ops.relattn_bwd_fused = orig_bwd, orig_fused
    kt = ops.ktime_collect() if not args.no_roofline else {}
    with torch.no_grad():
        loss = model(input_ids=ids, labels=labels).loss.item()
    out = None
    if rank == 0:
        tokens = B * T * world * args.steps
        d, L = cfg.d_model, cfg.n_layer
        f_fwd = flops_per_token_fwd(L, d, T, M, V)
        out = {
            'metric': 'train tokens/sec (TransfoXL seq2048 bf16), whole job', 'value': tokens / dt, 'unit': 'tokens/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'rccl_ranks': ranks.rccl_ranks(), 'grad_exchange_dtype': sync.dtype,
            'config': {'workload': wl['name'], 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': T, 'mem_len': M,
                       'parallelism': f'dp{world}', 'dropout': cfg.dropout, 'final_loss': loss,
                       'train_flops_per_token': 3 * f_fwd,
                       'whole_step_mfma_frac': 3 * f_fwd * tokens / dt / world / (MFMA_BF16_PEAK_TFLOPS * 1e12)},
        }
        if br.ev:
            ms = br.total_ms() / len(br.ev)
            # algorithmic flops of ONE attention-backward call (one layer, this rank's batch): backward = 2 x forward of the
            # banded attention core, forward = B*T*(4*d*n_bar + 2*d*M).  Split by the kernel that owns each gradient product:
            # query-owner dP + dQw (4 d n_bar) + dQr (2 d M), key-owner dK + dV (4 d n_bar), dRd contraction (2 d M); the
            # recomputed S / G products are not algorithmic work.
            nbar = (T + 1) / 2 if T <= M else M
            fwd_alg = B * T * (4 * d * nbar + 2 * d * M)
            alg = 2 * fwd_alg
            # zero-memory training: the dQr product over the all-phantom 256-distance blocks is formed by the FORWARD kernel (its
            # phantom value-sum, ops.relattn_fwd(..., oph=)) and enters the query-owner backward as an elementwise term: those
            # FLOPs are counted where they are executed, not in the backward group
            moved = 0.0
            fused = bool(getattr(eng._last, 'fused_bwd', False))
            Kc = eng._last.qkv[0].shape[1]
            if getattr(eng._last, 'oph', None) is not None:
                pz = -(((Kc - T) + 63) // 64) * 64
                if fused:       # every key position below the first stored one (oph_all)
                    cells = sum(max(0, M - 1 - (i - pz)) for i in range(T))
                else:           # the all-phantom 256-distance blocks
                    cells = sum(32 * max(0, M - (((g0 + 31 - pz) | 255) + 1)) for g0 in range(0, T, 32))
                moved = 2.0 * B * d * cells
            alg -= moved
            ach = alg / (ms * 1e-3) / 1e12
            traffic, src = pmc_traffic(args.workload, B)
            if fused:
                # per query: n_real stored keys in its window, M - n_real phantom distances.  The fused pass owns dP, dQw, dK, dV,
                # dQr and dRd over the stored keys (6 products); the phantom cells' dRd is the recompute kernel's; their dQr the forward's
                n_real = sum(min(M, i - (T - Kc) + 1) for i in range(T)) / T
                kalg = {'fwd': ('relattn_fwd_kernel', fwd_alg + moved), 'delta': ('fused_delta_kernel', 0.0),
                        'fused': ('relattn_bwd_fused_kernel', B * T * 12.0 * d * n_real), 'dqfin': ('relattn_dq_finish_kernel', 0.0),
                        'rowbias': ('phantom_prep_kernel', 0.0), 'drd': ('relattn_drd_phantom_kernel', B * T * 2.0 * d * (M - n_real))}
                group = ('delta', 'fused', 'dqfin', 'rowbias', 'drd')
                if 'delta' in kt and kt['delta'][1]:
                    desc = ('attention backward of one layer: fused_delta + relattn_bwd_fused + relattn_dq_finish + relattn_drd_phantom '
                            '(one HIP-event bracket around all four launches; the phantom kernel reads records the forward wrote)')
                else:       # round 6: delta = sum_e dO . O comes out of the epilogue of the GEMM that produces dO (mxl_gemm_bf16_headdot)
                    desc = ('attention backward of one layer: relattn_bwd_fused + relattn_dq_finish + relattn_drd_phantom (one HIP-event '
                            'bracket around the three launches; the row term delta is formed in the epilogue of the o_net input-gradient '
                            'GEMM in front of the bracket, +~16 us there instead of a 91 us pass; the phantom kernel reads records the '
                            'forward wrote)')
            else:
                kalg = {'fwd': ('relattn_fwd_kernel', fwd_alg + moved), 'delta': ('relattn_bwd_delta_kernel', 0.0),
                        'dq8': ('relattn_bwd_dq8_kernel', B * T * (4 * d * nbar + 2 * d * M) - moved),
                        'dkv': ('relattn_bwd_dkv_kernel', B * T * 4 * d * nbar), 'rowbias': ('add_rowbias_kernel', 0.0),
                        'drd': ('relattn_drd_kernel', B * T * 2 * d * M)}
                group = ('delta', 'dq8', 'dkv', 'rowbias', 'drd')
                desc = ('attention backward of one layer: relattn_bwd_delta + relattn_bwd_dq8 + relattn_bwd_dkv + add_rowbias + '
                        'relattn_drd (one HIP-event bracket around all five launches)')
            kernels = {}
            for key, (kname, kflop) in kalg.items():
                if key in kt and kt[key][1]:
                    kms = kt[key][0] / kt[key][1]
                    kernels[key] = {'kernel': kname, 'ms': kms, 'launches_timed': kt[key][1], 'alg_flop': kflop,
                                    'frac': kflop / (kms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
                    if key == 'fwd':      # on SURVEY 8(d)'s forward FLOPs alone (without the phantom dQr product it also forms)
                        kernels[key]['alg_flop_survey'] = fwd_alg
                        kernels[key]['frac_survey'] = fwd_alg / (kms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
            bwd_sum = sum(kernels[k]['ms'] for k in group if k in kernels)
            out['roofline'] = {'kernel': desc,
                               'bound': 'mfma', 'achieved': ach, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic,
                               'traffic_unit': 'HBM bytes per launch group (PMC)', 'traffic_source': src,
                               'avg_launch_ms': ms, 'launches_timed': len(br.ev), 'alg_flop_per_launch_group': alg,
                               'alg_flop_moved_to_forward': moved,
                               'sum_of_kernel_ms': bwd_sum, 'kernels': kernels}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_train(wl, T, M)
    del model, eng, sync
    torch.cuda.empty_cache()
    return out


def reformer_leg(args, ranks: Ranks, steps: int, warmup: int):
    """Reformer 6L (3 local + 3 LSH) d=512 H=8 dh=64 F=2048, T=8192 (axial 64x128), num_hashes=1, dropout on
    (0.05), auto num_buckets = [16,16].  Step = fwd + bwd + all-reduce + clip + AdamW."""
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    from symbolic_music_generation_amd.dist import GradSync
    from symbolic_music_generation_amd import ops
    dev, rank, world = ranks.dev, ranks.rank, ranks.world
    B = (args.batch if args.mode == 'reformer' and args.batch else 16)
    T = 8192
    cfg = MyReformerConfig('small', vocab_size=V, max_position_embeddings=T, axial_pos_shape=(64, 128), num_hashes=1)
    model = MyReformerModelWithLMHead(cfg, device=dev, seed=77).train()
    eng = model.engine
    sync = GradSync(eng)
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)
    ids = torch.randint(4, V, (B, T), generator=gen).to(dev)

    # dominant kernel of the step: the weight-gradient GEMM dW = dY^T X (split-K, fp32 atomics), 20 % of the kernel time
    br = EventBracket()
    orig_gemm = ops.gemm

    def timed_gemm(a, b, c, M_, N_, K_, **k):
        if br.on and k.get('trans_a') and (k.get('flags', 0) & ops.GEMM_OUT_F32_ATOMIC):
            return br.run(lambda: orig_gemm(a, b, c, M_, N_, K_, **k), work=2.0 * M_ * N_ * K_)
        return orig_gemm(a, b, c, M_, N_, K_, **k)

    if not args.no_roofline:
        ops.gemm = timed_gemm

    def step():
        with torch.no_grad():
            eng.zero_grad()
            model(input_ids=ids, labels=ids)
            eng.backward(layer_done=sync.layer_done)
            sync.finish()
            eng.optimizer_step(lr=3e-4, weight_decay=0.01, max_grad_norm=1.0, grad_scale=1.0 / world)

    def on_timed(on):
        br.on = on and not args.no_roofline
        if not args.no_roofline:
            ops.ktime_enable(on)

    dt = timed_steps(ranks, step, steps, warmup, on_timed=on_timed)
    ops.gemm = orig_gemm
    kt = ops.ktime_collect() if not args.no_roofline else {}
    out = None
    if rank == 0:
        d = cfg.hidden_size
        # local layer 24d^2 + 2*2*(2*64)*d, LSH layer 22d^2 + n_h*(d*rot + 512 d), head 2*2d*V; train = 3x
        rot = 32
        f_fwd = 3 * (24 * d * d + 512 * d) + 3 * (22 * d * d + (d * rot + 512 * d)) + 2 * 2 * d * V
        tokens = B * T * world * steps
        out = {'metric': 'train tokens/sec (Reformer 6L/512d seq8192 bf16), whole job', 'value': tokens / dt, 'unit': 'tokens/s',
               'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'dtype': 'bf16',
               'data': 'synthetic',
               'config': {'workload': 'SURVEY C4: Reformer small (3 local + 3 LSH) d=512 H8 dh64 F2048 T=8192 axial 64x128 n_h=1 '
                                      'V=1190, dropout 0.05', 'per_gpu_batch': B, 'train_flops_per_token': 3 * f_fwd,
                          'whole_step_mfma_frac': 3 * f_fwd * tokens / dt / world / (MFMA_BF16_PEAK_TFLOPS * 1e12)}}
        if br.ev:
            ms_total = br.total_ms()
            ach = br.work / (ms_total * 1e-3) / 1e12
            traffic, src = reformer_pmc_traffic(B)
            out['roofline'] = {'kernel': 'weight-gradient GEMM dW = dY^T X (gemm_tt256_kernel, or gemm_bf16_kernel<AT,BT> for shapes '
                                         'off the 256 grid; split-K fp32 atomics): every such launch of the timed steps',
                               'bound': 'mfma', 'achieved': ach, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic,
                               'traffic_unit': 'HBM bytes per launch (PMC), mean over the weight-gradient launches',
                               'traffic_source': src, 'avg_launch_ms': ms_total / len(br.ev), 'launches_timed': len(br.ev),
                               'share_of_step': ms_total * 1e-3 / dt}
        if kt.get('chunk_bwd_kv', (0, 0))[1]:
            # The chunked-attention kernels are gather-bound, not MFMA-bound: each is priced against its HBM floor.  Algorithmic
            # bytes per launch = every operand and result once, in units of one (B, T, d) bf16 tensor: forward q, k, v in + out;
            # query-owner backward q, k, v, dO, out in + dq out; key-owner backward q, k, v, dO, out in + dk, dv out.  In the LSH
            # layers q and k are one tensor (one unit less); three local and three LSH layers per step, so the mean is used.
            unit = B * T * d * 2.0
            units = {'chunk_fwd': ('chunk_attn_fwd_kernel', 3.5), 'chunk_bwd_q': ('chunk_attn_bwd_q_kernel', 5.5),
                     'chunk_bwd_kv': ('chunk_attn_bwd_kv_kernel', 6.5)}
            ck = {}
            for key, (kname, u) in units.items():
                if kt.get(key, (0, 0))[1]:
                    kms = kt[key][0] / kt[key][1]
                    ck[key] = {'kernel': kname, 'avg_launch_ms': kms, 'launches_timed': kt[key][1], 'algorithmic_bytes': u * unit,
                               'achieved': u * unit / (kms * 1e-3) / 1e9, 'frac': u * unit / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               'share_of_step': kt[key][0] * 1e-3 / dt}
            worst = min(ck, key=lambda k_: ck[k_]['frac'])
            out['roofline_hbm'] = {'kernel': ck[worst]['kernel'] + ': the kernel of this step furthest below its own roofline',
                                   'bound': 'hbm', 'achieved': ck[worst]['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': ck[worst]['frac'], 'traffic': None, 'avg_launch_ms': ck[worst]['avg_launch_ms'],
                                   'launches_timed': ck[worst]['launches_timed'], 'share_of_step': ck[worst]['share_of_step'],
                                   'algorithmic_bytes_per_launch': ck[worst]['algorithmic_bytes'], 'kernels': ck}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_reformer(T)
    del model, eng, sync
    torch.cuda.empty_cache()
    return out


def context_legs(args, ranks: Ranks):
    """Extra legs of the default one-GPU line (the headline fields are untouched): BASELINE.json configs[1]
    (C2), C3 with carried real memories (SURVEY 8(d) mode S, labelled), and the two configurations the reference's notebooks
    logged a throughput for (a single Tesla P100 with fp16 AMP: `vs_baseline` there is this run's tokens/s
    over that figure and says so -- different hardware, different precision, context only).  Each leg: a few untimed and at most
    five timed optimisation steps (forward + backward + clip + AdamW, dropout on), no roofline object, no CPU baseline."""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd.reformer import MyReformerConfig, MyReformerModelWithLMHead
    dev = ranks.dev
    steps, warmup = min(args.steps, 5), 2
    legs = {}

    def run(name, model, ids, flops_tok, workload, mems=None, extra=None):
        eng = model.engine

        def step():
            with torch.no_grad():
                eng.zero_grad()
                if mems is not None:
                    eng.forward(ids, mems=mems, labels=ids, train=True)
                else:
                    model(input_ids=ids, labels=ids)
                eng.backward()
                eng.optimizer_step(lr=3e-4, weight_decay=0.01, max_grad_norm=1.0)

        dt = timed_steps(ranks, step, steps, warmup)
        toks = ids.numel() * steps
        legs[name] = {'metric': 'train tokens/sec, one MI355X', 'value': toks / dt, 'unit': 'tokens/s', 'steps': steps, 'warmup': warmup,
                      'ms_per_step': 1e3 * dt / steps, 'dtype': 'bf16', 'data': 'synthetic', 'vs_baseline': None,
                      'config': {'workload': workload, 'per_gpu_batch': ids.shape[0], 'seq_len': ids.shape[1],
                                 'train_flops_per_token': flops_tok,
                                 'whole_step_mfma_frac': flops_tok * toks / dt / (MFMA_BF16_PEAK_TFLOPS * 1e12)}}
        if extra:
            legs[name].update(extra)

    gen = torch.Generator(device='cpu').manual_seed(77)
    # ---- C2 (BASELINE.json configs[1]), mode R
    wl = WORKLOADS['c2']
    cfg = MyTransfoXLConfig(wl['size'], max_length=wl['T'], vocab_size=V, n_layer=wl['n_layer'], mem_len=wl['M'], cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    ids = torch.randint(4, V, (wl['B'], wl['T']), generator=gen).to(dev)
    run('c2', model, ids, 3 * flops_per_token_fwd(cfg.n_layer, cfg.d_model, wl['T'], wl['M'], V), wl['name'])
    del model
    torch.cuda.empty_cache()
    # ---- C3, mode S: the second 2048-token segment of a stream, with the first segment's (detached) memories carried in
    wl = WORKLOADS['c3']
    cfg = MyTransfoXLConfig(wl['size'], max_length=wl['T'], vocab_size=V, n_layer=wl['n_layer'], mem_len=wl['M'], cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    Bs = 32
    seg0 = torch.randint(4, V, (Bs, wl['T']), generator=gen).to(dev)
    ids = torch.randint(4, V, (Bs, wl['T']), generator=gen).to(dev)
    with torch.no_grad():
        mems = [m.detach() for m in model.engine.forward(seg0, train=False)['mems']]
    d, L, T, M = cfg.d_model, cfg.n_layer, wl['T'], wl['M']
    f_fwd_s = L * (24 * d * d + 6 * d * M) + 2 * d * V             # SURVEY 8(d) mode S: n_bar = M, attention 6 d M
    run('c3_mode_s', model, ids, 3 * f_fwd_s,
        'TransfoXL 12L/768d T=2048 M=2048 V=1190, SURVEY 8(d) MODE S: segment-recurrent, carried real memories (Kc = 4096); '
        'the headline line is mode R', mems=mems)
    del model, mems
    torch.cuda.empty_cache()
    # ---- the reference's own logged TransfoXL run: base, seq 512, mem 256, batch 32 (notebook/train/transformer-xl.ipynb:491-698)
    cfg = MyTransfoXLConfig('base', max_length=512, vocab_size=418, mem_len=256, cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).train()
    ids = torch.randint(4, 418, (32, 512), generator=gen).to(dev)
    run('published_transfoxl_base', model, ids, 3 * flops_per_token_fwd(cfg.n_layer, cfg.d_model, 512, 256, 418),
        'TransfoXL base 12L/768d seq 512 mem 256 V=418 batch 32, mode R (the run logged in notebook/train/transformer-xl.ipynb)')
    legs['published_transfoxl_base'].update(
        vs_baseline=legs['published_transfoxl_base']['value'] / 2815.0,
        baseline={'value': 2815.0, 'unit': 'tokens/s', 'hardware': '1x Tesla P100-PCIE-16GB, fp16 AMP (DIFFERENT HARDWARE: context only)',
                  'source': 'BASELINE.md section 1; notebook/train/transformer-xl.ipynb:491,698 (epoch wall time)'})
    del model
    torch.cuda.empty_cache()
    # ---- the reference's own logged Reformer run: base, seq 4096, num_hashes 2, batch 17 (notebook/train/reformer.ipynb:2870)
    cfg = MyReformerConfig('base', vocab_size=420, max_position_embeddings=4096, axial_pos_shape=(64, 64))
    model = MyReformerModelWithLMHead(cfg, device=dev, seed=77).train()
    ids = torch.randint(4, 420, (17, 4096), generator=gen).to(dev)
    dr, nh = cfg.hidden_size, cfg.num_hashes
    nl = len(cfg.attn_layers) // 2
    f_fwd_r = nl * (24 * dr * dr + 512 * dr) + nl * (22 * dr * dr + nh * (dr * 128 + 512 * dr)) + 2 * 2 * dr * 420
    run('published_reformer_base', model, ids, 3 * f_fwd_r,
        f'Reformer base {len(cfg.attn_layers)}L/768d seq 4096 num_hashes={nh} V=420 batch 17 (the run logged in notebook/train/reformer.ipynb)')
    legs['published_reformer_base'].update(
        vs_baseline=legs['published_reformer_base']['value'] / 5849.0,
        baseline={'value': 5849.0, 'unit': 'tokens/s', 'hardware': '1x Tesla P100-PCIE-16GB, fp16 AMP (DIFFERENT HARDWARE: context only)',
                  'source': 'BASELINE.md section 1; notebook/train/reformer.ipynb:2870-2871 (train_samples_per_second 1.428 x 4096)'})
    del model
    torch.cuda.empty_cache()
    return legs


def decode_leg(args, ranks: Ranks, warmup: int):
    """TransfoXL 12L/768d cached-mem decode, batch 64 prompts x 256 tokens, top-k 8, generate to 2048,
    one hipGraph replay per token.  A 'step' here = one generated token for the whole batch."""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd.generate import XLDecoder, XLDecoderLanes
    dev, rank = ranks.dev, ranks.rank
    B, Tp, M = (args.batch if args.mode == 'decode' and args.batch else 64), args.decode_prompt, 2048
    cfg = MyTransfoXLConfig('base', max_length=2048, vocab_size=V, mem_len=M, cutoffs=[])
    model = MyTransfoXLLMHeadModel(cfg, device=dev, seed=77).eval()
    # room past T = 2048 so that a full-ring window can be timed after the C5 generation proper
    FULL = 128
    # the decoder model.generate builds for this batch: two free-running half-batch lanes from 32 rows on (generate.XLDecoderLanes;
    # MXL_DECODE_LANES=1: one decoder).  The eager (PMC) mode keeps one decoder on one stream.
    lanes = int(os.environ.get('MXL_DECODE_LANES', '2')) if (B >= 32 and not args.eager) else 1
    dec = (XLDecoderLanes(model.engine, B, 2048 + FULL + 8, seed=77 + rank, lanes=lanes) if lanes > 1
           else XLDecoder(model.engine, B, 2048 + FULL + 8, seed=77 + rank))
    gen = torch.Generator(device='cpu').manual_seed(77 + rank)
    prompt = torch.randint(4, V, (B, Tp), generator=gen).to(dev)
    samp = dict(do_sample=True, top_k=8, top_p=1.0, temperature=1.0, repetition_penalty=1.0, typical_p=1.0)
    warm = max(warmup, 1)
    with torch.no_grad():
        dec.begin(prompt, 2048 + FULL + 8, samp, use_graph=not args.eager)     # prompt pass, first token, graph capture
        for _ in range(warm):
            dec.replay_once()
        replay = dec.replay_once
        done = Tp + 1 + warm              # positions filled so far: prompt, its sample, warm-up steps
        torch.cuda.synchronize()
        # the whole C5 generation, prompt 256 -> T = 2048 (the first steps are cheaper than the last: ring slots that were never
        # written are HF's zero mems and cost no K/V bytes, so a short window after the prompt would flatter the number)
        steps = 2048 - done if args.decode_steps <= 0 else min(args.decode_steps, 2048 - done)
        t0 = time.perf_counter()
        for _ in range(steps):
            replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # steady state with every ring slot written: positions 2048 .. 2048 + FULL
        full_steps = FULL if args.decode_steps <= 0 else 0
        dt_full = None
        if full_steps:
            t0 = time.perf_counter()
            for _ in range(full_steps):
                replay()
            torch.cuda.synchronize()
            dt_full = time.perf_counter() - t0
    d, L = cfg.d_model, cfg.n_layer
    # algorithmic bytes per step: weights once + the WRITTEN part of the projected K/V ring per sequence, averaged
    # over the timed steps; `full_ring` is the figure for a full memory (2 M d 2 B per layer)
    valid = sum(min(done + i, M) for i in range(steps)) / max(steps, 1)
    wbytes = L * 12 * d * d * 2 + V * d * 2
    bytes_step = wbytes + B * L * 2 * valid * d * 2
    bytes_full = wbytes + B * L * 2 * M * d * 2
    out = None
    if rank == 0:
        ach = bytes_step / (dt / steps) / 1e9
        out = {'metric': 'AR decode tokens/sec (TransfoXL 12L/768d, batch 64, cached mems, top-k 8, hipGraph step)',
               'value': B * steps / dt, 'unit': 'tokens/s', 'n_gpus': 1, 'steps': steps, 'warmup': warm,
               'ms_per_step': 1e3 * dt / steps, 'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': 'SURVEY C5 decode: 12L/768d, M=2048, B=64 prompts x 256 tokens generated to T=2048, top_k=8',
                          'batch': B, 'positions_timed': [done, done + steps], 'hipgraph': not args.eager, 'lanes': lanes},
               'roofline': {'kernel': 'whole decode step (one hipGraph replay: 12 x [qkv+append, bd, ring attention, o, LN, ffn1, '
                                      'ffn2 slabs, LN] + head + sampler)', 'bound': 'hbm', 'achieved': ach,
                            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, **decode_pmc_traffic(),
                            'basis': 'written ring slots, averaged over the timed generation',
                            'algorithmic_bytes_per_step': bytes_step, 'mean_valid_ring_slots': valid}}
        if dt_full:
            ach_f = bytes_full / (dt_full / full_steps) / 1e9
            out['full_ring'] = {'value': B * full_steps / dt_full, 'unit': 'tokens/s', 'steps': full_steps,
                                'ms_per_step': 1e3 * dt_full / full_steps, 'positions_timed': [2048, 2048 + full_steps],
                                'roofline': {'bound': 'hbm', 'achieved': ach_f, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                             'frac': ach_f / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': bytes_full}}
        if ranks.world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_decode(Tp, M)
    del dec, model
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ PMC traffic (recorded)
def pmc_traffic(workload, B):
    """HBM bytes per attention-backward group from the newest committed PMC passes (profiles/r*_c3_pmc_traffic.json, produced
    by scripts/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE rocprofv3 runs of this same command; FETCH doubled per
    the gfx950 correction).  Counters cannot be collected from inside the timed run, so this is the recorded measurement for
    the default workload and None for any other batch / workload / kernel set."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_c3_pmc_traffic.json')))
    if workload != 'c3' or not files:
        return None, None
    path = files[-1]
    rec = json.load(open(path))
    if rec.get('per_gpu_batch', 16) != B:
        return None, None
    k = rec['kernels']
    names = rec.get('attention_backward_group')
    if names is None:       # every launch of the bracketed group
        fused = [n for n in k if n.startswith('relattn_bwd_fused_kernel')]
        if fused:           # round 4: delta, the fused pass, the dq finish, q + r_r_bias (+ per-tile scalars), the phantom cells' dRd
            if 'relattn_drd_phantom_kernel' in k:       # (the phantom kernel's records come out of the forward: no q + r_r_bias pass)
                names = ['fused_delta_kernel', fused[0], 'relattn_dq_finish_kernel', 'relattn_drd_phantom_kernel']
            else:
                names = ['fused_delta_kernel', fused[0], 'relattn_dq_finish_kernel', 'add_rowbias_kernel', 'relattn_drd_kernel']
        else:               # delta, query-owner, key-owner, q + r_r_bias, dRd contraction
            dq = 'relattn_bwd_dq8_kernel<64>' if 'relattn_bwd_dq8_kernel<64>' in k else 'relattn_bwd_dq_kernel<64>'
            names = ['relattn_bwd_delta_kernel', dq, 'relattn_bwd_dkv_kernel<64>', 'add_rowbias_kernel', 'relattn_drd_kernel']
    # round 6: delta comes out of the o_net input-gradient GEMM's epilogue (in front of the bracket): no fused_delta_kernel launch then
    names = [n for n in names if n != 'fused_delta_kernel' or n in k]
    if not all(n in k for n in names):
        return None, None
    if 'group_sources_sha16' in rec:      # the record is only as good as the kernels it measured: a changed source needs new PMC passes
        sys.path.insert(0, os.path.join(ROOT, 'scripts'))
        import pmc_traffic as _pt
        if _pt.sources_sha16('train') != rec['group_sources_sha16']:
            return None, os.path.basename(path) + ' (STALE: kernel sources changed since the PMC passes)'
    return sum(k[n]['hbm_bytes_per_launch'] * k[n].get('launches_per_group', 1) for n in names), os.path.basename(path)


def reformer_pmc_traffic(B):
    """HBM bytes per weight-gradient GEMM launch of the C4 leg from the newest committed PMC passes
    (profiles/r*_c4_pmc_traffic.json: scripts/collect_profiles.sh, separate FETCH_SIZE / WRITE_SIZE runs of
    `bench.py --mode reformer`), averaged over the launches of the kernels that serve those GEMMs"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_c4_pmc_traffic.json')))
    if not files:
        return None, None
    rec = json.load(open(files[-1]))
    if rec.get('per_gpu_batch') != B:
        return None, None
    ks = {n: v for n, v in rec['kernels'].items() if n.startswith('gemm_tt256_kernel') or n.startswith('gemm_bf16_kernel<true, true')}
    tot = sum(v['launches'] for v in ks.values())
    if not tot:
        return None, None
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks.values()) / tot, os.path.basename(files[-1])


def decode_pmc_traffic():
    """HBM bytes per decode step from the newest committed PMC passes of an EAGER decode window (profiles/r*_c5_decode_eager_
    pmc_traffic.json: rocprofv3 cannot collect counters over hipGraph replays; same kernels, launched one by one, over the ring
    positions stated in the file)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_c5_decode_eager_pmc_traffic.json')))
    if not files:
        return {'traffic': None}
    rec = json.load(open(files[-1]))
    return {'traffic': rec.get('hbm_bytes_per_decode_step'), 'traffic_unit': 'HBM bytes per decode step (PMC, eager launches)',
            'traffic_source': os.path.basename(files[-1]), 'traffic_positions': rec.get('positions'),
            'traffic_algorithmic_bytes_at_those_positions': rec.get('algorithmic_bytes_per_step_at_these_positions')}


# ------------------------------------------------------------------------------------------------ CPU oracle baselines
def _timed_cpu_steps(fn, max_steps=3, budget=CPU_BUDGET_S):
    """1 warm-up + up to `max_steps` timed calls, stopping early once `budget` seconds of CPU work are spent"""
    t_all = time.perf_counter()
    fn()
    times = []
    while len(times) < max_steps and (not times or time.perf_counter() - t_all + times[-1] < budget):
        t = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t)
    return times


def cpu_baseline_train(wl, T, M):
    """The oracle (reference-style dense fp32 TransfoXL, oracle/transfoxl_ref.py) on the host cores: train steps (fwd + bwd +
    clip + AdamW), B = 1 sequence of the same T / M.  A full 12-layer step takes 80-140 s on the box's host, so the sample is
    the embedding / head part alone and models of 1 and 2 decoder layers (1 warm-up + 1-2 timed steps each); the full step is
    t(head part) + L x (the last layer increment timed), with the range over all increments reported (`extrapolation_range`)."""
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    torch.manual_seed(77)
    L_full = wl['n_layer']
    ids = torch.randint(4, V, (1, T))

    def make(L):
        c = RefXLConfig.from_preset(wl['size'], vocab_size=V, max_length=T, mem_len=M, cutoffs=[], n_layer=L)
        m = RefTransfoXLLMHeadModel(c).train()
        opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=0.1)

        def one():
            o = m(ids, labels=ids)
            o.loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
        return one, m, opt

    one, m, opt = make(1)

    def head_part():
        """embedding + adaptive-softmax head + loss of the same model, without the decoder layer"""
        h = m.transformer.word_emb(ids.transpose(0, 1)).transpose(0, 1)
        nll = m.crit(h, ids)
        nll[nll != 0].mean().backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad()

    t0 = _timed_cpu_steps(head_part, 3, 5.0)
    t1 = _timed_cpu_steps(one, 2, CPU_BUDGET_S)
    a, b = sum(t0) / len(t0), sum(t1) / len(t1)
    times = {0: a, 1: b}                                  # layers -> seconds per step
    sample = (f'B=1 x T={T} (M={M}) fp32 train steps (fwd+bwd+clip+AdamW), each model after 1 warm-up step: embedding+head part '
              f'alone {a:.2f} s ({len(t0)} steps), 1-layer model {b:.2f} s ({len(t1)} steps)')
    if L_full > 1:
        # A 12-layer step is ~80-140 s and does not fit the bench's time budget, so the per-layer cost comes from models of 1, 2
        # (and with MXL_CPU_BASELINE_LAYERS=3, 3) layers.  Layers are identical in arithmetic but not in cost: on some hosts the
        # second layer of a 2-layer model costs 1.6 x the first (cache / NUMA footprint of the dense (T, T + M) score tensors),
        # so the step is extrapolated from the LAST increment timed -- the one that already pays for a neighbour layer, as ten of
        # the twelve layers do -- and the range over all increments is reported beside it.
        del one, m, opt
        for L in range(2, max(2, int(os.environ.get('MXL_CPU_BASELINE_LAYERS', '2'))) + 1):
            fn, mL, optL = make(L)
            tL = _timed_cpu_steps(fn, 1, 2.5 * CPU_BUDGET_S)
            times[L] = sum(tL) / len(tL)
            sample += f', {L}-layer model {times[L]:.2f} s'
            del fn, mL, optL
    Ls = sorted(times)
    incr = [max(times[Ls[i + 1]] - times[Ls[i]], 1e-9) for i in range(len(Ls) - 1)]      # cost of layer 1, 2, ...
    per_layer = incr[-1]
    t_full = a + L_full * per_layer
    lo, hi = a + L_full * max(incr), a + L_full * min(incr)
    out = {'value': T / t_full, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'extrapolated': True, 'layers_timed': Ls[-1], 'layers_full': L_full,
           'per_layer_s': [round(x, 3) for x in incr],
           'extrapolation_range': [round(T / lo, 3), round(T / hi, 3)],
           'sample': sample + f'; full step = head part + {L_full} x the last layer increment ({per_layer:.2f} s) = {t_full:.1f} s'}
    if len(incr) > 1:
        out['second_layer_over_first'] = round(incr[1] / incr[0], 3)
    return out


def cpu_baseline_decode(Tp, M):
    """oracle greedy decode (HF-style loop: one forward per token over cat(mems, token), mems carried) at the C5 model,
    B = 1, 256-token prompt, 32 generated tokens"""
    from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
    torch.manual_seed(77)
    c = RefXLConfig.from_preset('base', vocab_size=V, max_length=2048, mem_len=M, cutoffs=[])
    m = RefTransfoXLLMHeadModel(c).eval()
    ids = torch.randint(4, V, (1, Tp))
    n_new = 32
    with torch.no_grad():
        t = time.perf_counter()
        out = m(ids)                                   # prompt pass (not counted in tokens/s: the GPU number excludes it too)
        t_prompt = time.perf_counter() - t
        past, cur = out.mems, out.prediction_scores[:, -1].argmax(-1, keepdim=True)
        t = time.perf_counter()
        for _ in range(n_new):
            o = m(cur, mems=past)
            past, cur = o.mems, o.prediction_scores[:, -1].argmax(-1, keepdim=True)
        dt = time.perf_counter() - t
    return {'value': n_new / dt, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'B=1, {Tp}-token prompt ({t_prompt:.2f} s, not counted), {n_new} greedy tokens with carried mems '
                      f'(M={M}), fp32, {dt:.2f} s'}


def cpu_baseline_reformer(T):
    """oracle/reformer_ref.py (pinned on HF Reformer goldens) at C4, B = 1: forward + loss + autograd backward (stored
    activations; HF's reversible stack would recompute every layer on top of this) + clip + AdamW"""
    from oracle.reformer_ref import RefReformerConfig, RefReformer, init_params
    torch.manual_seed(77)
    c = RefReformerConfig.from_preset('small', vocab_size=V, max_position_embeddings=T, axial_pos_shape=(64, 128), num_hashes=1,
                                      hidden_dropout_prob=0.0, local_attention_probs_dropout_prob=0.0)
    params = {k: v.requires_grad_(True) for k, v in init_params(c, seed=77).items()}
    m = RefReformer(c, params)
    opt = torch.optim.AdamW(list(params.values()), lr=3e-4, weight_decay=0.01)
    ids = torch.randint(4, V, (1, T))
    g = torch.Generator().manual_seed(77)
    rots = {l: torch.randn(*m.rotations_shape(T), generator=g) for l, kind in enumerate(c.attn_layers) if kind == 'lsh'}

    def one():
        _, loss = m.forward(ids, rotations=rots, labels=ids)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        opt.zero_grad()

    times = _timed_cpu_steps(one, 3, CPU_BUDGET_S)
    t = sum(times) / len(times)
    return {'value': T / t, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'B=1 x T={T} fp32 train steps (fwd+bwd+clip+AdamW, no reversible recompute) after 1 warm-up: {len(times)} '
                      f'steps, {t:.2f} s each'}


# ------------------------------------------------------------------------------------------------ CPU stub (tests only)
def stub_leg(args, ranks: Ranks):
    """`MXL_BENCH_STUB=1`: no GPU, gloo ranks, a step = one all-reduce of a small buffer.  Exists so that a CPU test can run
    `bench.py --gpus 2` through the real launcher / barrier / max-over-ranks / relay code path."""
    buf = torch.ones(1024)

    def step():
        ranks.dist.all_reduce(buf) if ranks.dist is not None else None
        buf.fill_(1.0)

    dt = timed_steps(ranks, step, args.steps, args.warmup)
    if ranks.rank == 0:
        return {'metric': 'stub', 'value': args.steps / dt, 'unit': 'steps/s', 'n_gpus': ranks.world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'rccl_ranks': ranks.rccl_ranks(), 'stub': True}
    return None


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    launched = 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not launched:
        raise SystemExit(spawn(args, argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if launched and world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if os.environ.get('MXL_BENCH_STUB') == '1':
        ranks = Ranks('gloo', torch.device('cpu'))
        out = stub_leg(args, ranks)
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local_rank)
        ranks = Ranks('nccl', torch.device('cuda', local_rank))
        if args.mode == 'decode':
            out = decode_leg(args, ranks, args.warmup)
        elif args.mode == 'reformer':
            out = reformer_leg(args, ranks, args.steps, args.warmup)
            if out is not None:
                out.update(higher_is_better=True, scaling='weak', vs_baseline=None, rccl_ranks=ranks.rccl_ranks())
        else:
            out = train_leg(args, ranks)
            if args.mode == 'all' and world == 1 and args.workload == 'c3':
                dec = decode_leg(args, ranks, 3)
                ref = reformer_leg(args, ranks, min(args.steps, 10), min(args.warmup, 3))
                out['decode'], out['reformer'] = dec, ref
                if not args.no_context_legs:
                    out['context'] = context_legs(args, ranks)
    if out is not None:
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == '__main__':
    main()
