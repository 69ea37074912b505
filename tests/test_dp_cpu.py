"""Data-parallel host logic on CPU: `bench.py --gpus N` really starts N ranks (gloo stub), every rank yields the same number
of equally sized training batches whatever the dataset size, and 2 ranks x B sequences give the gradients and loss of 1 rank
x 2B through `dist.GradSync` (the oracle model supplies real tiny-model gradients; the HIP engine cannot run here)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port(k=0):
    return 29700 + (os.getpid() * 7 + k * 131) % 2000


def test_bench_gpus_flag_starts_that_many_ranks():
    env = dict(os.environ, MXL_BENCH_STUB='1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--master-port', str(_port())], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0 only)'
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['steps'] == 3 and out['warmup'] == 1


def test_bench_spawn_argv_and_world_mismatch():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(['--gpus', '8', '--steps', '20', '--warmup', '5'])
    argv = bench.spawn_argv(args, ['--gpus', '8', '--steps', '20', '--warmup', '5'])
    assert argv[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in argv
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1'
    assert argv[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5'] and argv[-7].endswith('bench.py')
    # started by a launcher with a different world size: fail loudly instead of reporting the wrong n_gpus
    env = dict(os.environ, MXL_BENCH_STUB='1', WORLD_SIZE='2', RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stdout + r.stderr)


@pytest.mark.parametrize('n,bsz', [(127, 32), (130, 32), (64, 32), (5, 2)])
def test_every_rank_yields_the_same_batches(monkeypatch, n, bsz):
    """n % world != 0 gave the ranks different step counts (n=127, world=2, bsz=32: 2 vs 1), which misaligns the
    per-layer gradient all-reduces.  Now the shard is padded DistributedSampler-style."""
    from symbolic_music_generation_amd import trainer as tr, dist as mdist
    from symbolic_music_generation_amd.data import DeviceBatcher
    world = 2
    ds = [torch.full((4,), i) for i in range(n)]
    shapes, seen = [], []
    for rank in range(world):
        monkeypatch.setattr(mdist, 'world_size', lambda: world)
        monkeypatch.setattr(mdist, 'rank', lambda r=rank: r)
        t = object.__new__(tr.MyTrainer)
        t.seed = 77

        class M:
            device = 'cpu'
        t.model = M()
        bs = list(t._batches(ds, bsz, epoch=3, shuffle=True, pad=True))
        shapes.append([b.shape[0] for b in bs])
        seen += [int(b[j, 0]) for b in bs for j in range(b.shape[0])]
        # evaluation: plain strided shard, no duplicates
        ev = [i for b, rows in t._batches(ds, bsz, 0, shuffle=False, with_index=True) for i in rows]
        assert ev == list(range(n))[rank::world]
        db = object.__new__(DeviceBatcher)
        db.tf, db.shuffle, db.seed, db.epoch, db.rank, db.world, db.B, db.drop_last = ds, True, 1, 0, rank, world, bsz, False
        shapes.append(('db', len(db.order()), len(db)))
    assert shapes[0] == shapes[2] and shapes[1][1:] == shapes[3][1:], shapes
    assert set(seen) == set(range(n)) and len(seen) == (n + world - 1) // world * world
    assert len(shapes[0]) == -(-((n + world - 1) // world) // bsz)


_DP_EQUIV_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from symbolic_music_generation_amd.dist import GradSync
from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
from symbolic_music_generation_amd.xl_engine import ParamLayout
from oracle.transfoxl_ref import RefXLConfig, RefTransfoXLLMHeadModel
world = 2
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2], rank=int(sys.argv[3]), world_size=world)
rank = dist.get_rank()
kw = dict(vocab_size=1190, cutoffs=[], n_layer=2, mem_len=32, max_length=32, dropout=0.0)
cfg = MyTransfoXLConfig('debug', **kw)
layout = ParamLayout(cfg)
torch.manual_seed(5)
ref = RefTransfoXLLMHeadModel(RefXLConfig.from_preset('debug', **kw)).train()
ids = torch.randint(4, 1190, (4, 32), generator=torch.Generator().manual_seed(9))

def flat_grads(batch):
    ref.zero_grad()
    out = ref(batch, labels=batch)
    out.loss.backward()
    G = torch.zeros(layout.total)
    for n, p in ref.named_parameters():
        if n in layout.entries:
            layout.view(G, n).copy_(p.grad)
    return G, out.loss.item()

class E: pass
e = E(); e.cfg, e.layout = cfg, layout
e.G, loss = flat_grads(ids[rank * 2:(rank + 1) * 2])          # this rank's shard: B = 2
gs = GradSync(e)
for l in reversed(range(cfg.n_layer)): gs.layer_done(l)
gs.finish()
g_dp = e.G / world                                            # the 1/world the fused AdamW folds in (grad_scale)
l = torch.tensor([loss]); dist.all_reduce(l); loss_dp = l.item() / world
g_full, loss_full = flat_grads(ids)                           # one rank, 2B = 4
assert abs(loss_dp - loss_full) < 1e-5 * abs(loss_full), (loss_dp, loss_full)
err = ((g_dp - g_full).norm() / g_full.norm()).item()
assert err < 1e-5, err
assert g_full.abs().sum() > 0
dist.destroy_process_group()
print('ok', err)
'''


def test_two_ranks_times_B_equals_one_rank_times_2B(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_DP_EQUIV_WORKER)
    port = str(_port(1))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('ok' in o for o in outs)


def test_bf16_exchange_error_bound():
    """The opt-in bf16 gradient exchange (`GradSync(engine, 'bf16')`): every rank's bucket is rounded to bf16 (8
    significant bits) and the 8-rank ring sums in bf16, rounding each partial sum.  Emulated here on gradient-like data (a common
    signal plus per-rank noise, as data-parallel replicas see) against the fp32 sum: every element within
    (world + 1) * 2^-8 * sum_r |g_r| (the worst case of the roundings), the whole vector within 1 % rel-Frobenius -- and the fp32
    exchange (the default) is exact to fp32 summation order."""
    torch.manual_seed(0)
    world, n = 8, 1 << 18
    common = torch.randn(n) * torch.logspace(-6, -2, n)              # gradients span orders of magnitude
    g = [common + 0.5 * common.abs() * torch.randn(n) for _ in range(world)]
    exact = torch.stack(g).double().sum(0)
    # ring reduce-scatter order for one chunk: rank r adds its bf16 value to the bf16 running sum it received
    run = g[0].to(torch.bfloat16)
    for r in range(1, world):
        run = (run.float() + g[r].to(torch.bfloat16).float()).to(torch.bfloat16)
    got = run.double()
    bound = (world + 1) * 2.0 ** -8 * torch.stack(g).double().abs().sum(0)
    assert ((got - exact).abs() <= bound + 1e-30).all()
    rel = ((got - exact).norm() / exact.norm()).item()
    assert rel < 1e-2, rel
    f32 = torch.stack(g).sum(0).double()
    assert ((f32 - exact).norm() / exact.norm()).item() < 1e-6
    # the default exchange dtype is fp32 (what HF's DDP exchanges in the reference stack)
    from symbolic_music_generation_amd.dist import GradSync
    import inspect
    assert "'fp32'" in inspect.getsource(GradSync.__init__)


def _check_partition(layout, n_layer):
    """layer buckets + rest cover [0, layout.total) exactly once: no gradient element is exchanged twice or left out"""
    from symbolic_music_generation_amd.dist import layer_buckets
    per_layer, rest = layer_buckets(layout, n_layer)
    assert len(per_layer) == n_layer
    hits = np.zeros(layout.total, dtype=np.int32)
    for sl in per_layer:
        for lo, hi in sl:
            assert 0 <= lo < hi <= layout.total
            hits[lo:hi] += 1
    for lo, hi in rest:
        assert 0 <= lo < hi <= layout.total
        hits[lo:hi] += 1
    assert hits.min() == 1 and hits.max() == 1, f'covered {int((hits == 1).sum())} of {layout.total} once, max {hits.max()}'
    # every decay-segment parameter of layer l lies in layer l's ONE bucket (so its exchange can start when that layer's backward
    # is enqueued); the layers' no-decay parameters (LayerNorm, biases: a few KB each) travel together in the tail
    prefix = getattr(layout, 'layer_prefix', lambda l: f'transformer.layers.{l}.')
    for l in range(n_layer):
        assert len(per_layer[l]) == 1
        for name, (off, shape) in layout.entries.items():
            if name.startswith(prefix(l)):
                n = int(np.prod(shape))
                if off < layout.n_decay:
                    assert any(lo <= off and off + n <= hi for lo, hi in per_layer[l]), name
                else:
                    assert any(lo <= off and off + n <= hi for lo, hi in rest), name
    assert len(rest) <= 3, f'{len(rest)} tail messages'


def test_gradient_buckets_partition_the_flat_buffer():
    """the per-layer all-reduce buckets of dist.GradSync and the head / embedding remainder partition the flat
    gradient buffer exactly once -- for the Transformer-XL engine (the reference's vocabularies, no cutoffs and [1000]), for the
    bucketed large-vocabulary head ([10000] and [20000, 40000, 200000]: `transformer_xl.py:53-66`) and for the Reformer engine."""
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
    from symbolic_music_generation_amd.xl_engine import ParamLayout
    from symbolic_music_generation_amd.reformer import MyReformerConfig
    from symbolic_music_generation_amd.rf_engine import RFLayout

    class Tok:
        def __init__(self, v):
            self.vocab_size = v
            self.pad_token_id, self.eos_token_id, self.model_max_length = 1, 3, 2048

        def __len__(self):
            return self.vocab_size

    for size, V, kw in [('base', 1190, dict(cutoffs=[])), ('base', 1190, {}), ('small', 32768, {}), ('debug', 262144, {}),
                        ('tiny', 422, dict(n_layer=3))]:
        cfg = MyTransfoXLConfig(model_size=size, tokenizer=Tok(V), **kw)
        _check_partition(ParamLayout(cfg), cfg.n_layer)
    for size in ('debug', 'small', 'base'):
        cfg = MyReformerConfig(model_size=size, tokenizer=Tok(420))
        _check_partition(RFLayout(cfg), len(cfg.attn_layers))


def test_bench_gpus_flag_at_eight_ranks():
    """the driver's N = 8 launch shape, on gloo: eight ranks start, barrier, take the max over ranks, one JSON line from rank 0"""
    env = dict(os.environ, MXL_BENCH_STUB='1', OMP_NUM_THREADS='1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1',
                        '--master-port', str(_port(3))], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0 only)'
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['rccl_ranks'] == 8 and out['steps'] == 3
