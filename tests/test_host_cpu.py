"""CPU-side tests: C-ABI surface, host logic (config presets, parameter layout, vocabulary, schedules), and the
data-parallel gradient exchange over gloo with world_size 2.  No GPU compute."""
import ctypes
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    from symbolic_music_generation_amd import _lib
    decl = _lib.declared_functions()
    assert len(decl) >= 25 and 'mxl_relattn_fwd' in decl and 'mxl_gemm_bf16' in decl
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f'{name} declared in include/musicxl.h but not exported'
    assert _lib.lib().mxl_abi_version() == 1
    assert b'invalid argument' in _lib.lib().mxl_error_string(-1)


def test_cabi_argument_errors_without_gpu():
    """argument validation happens before any launch: callable on a CPU-only box."""
    from symbolic_music_generation_amd import _lib
    L = _lib.lib()
    assert L.mxl_gemm_bf16(None, None, None, 8, 8, 8, 8, 8, 8, 0, 0, 0, 1.0, None, None, 0, 1, 0.0, 0, 0, None) == -1
    assert L.mxl_relattn_fwd(*([None] * 8), 1, 1, 1, 64, 64, 1, 0, 0, 0, 0, 0, 0, 0, 1.0, None) == -1


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig, MyTransfoXLLMHeadModel
    from symbolic_music_generation_amd._lib import MusicXLError
    with pytest.raises(MusicXLError):
        MyTransfoXLLMHeadModel(MyTransfoXLConfig('debug', vocab_size=100, cutoffs=[]))


def test_no_oracle_import_in_product_package():
    pkg = os.path.join(ROOT, 'symbolic_music_generation_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py') and fn != 'smoke.py':
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f'{fn} imports the oracle'


def test_config_presets_match_reference():
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
    from symbolic_music_generation_amd.vocab import MusicTokenizer
    c = MyTransfoXLConfig('base')
    assert (c.d_model, c.n_head, c.n_layer, c.d_head, c.d_inner, c.mem_len, c.clamp_len) == (768, 12, 12, 64, 3072, 256, 1024)
    assert c.max_length_ == 2048 and c.same_length and c.div_val == 1 and c.eos_token_id == 0
    d = MyTransfoXLConfig('debug')
    assert (d.d_model, d.d_head, d.mem_len, d.clamp_len, d.max_length_) == (128, 16, 64, 64, 64)
    tok = MusicTokenizer(pitch_kind='degree')
    assert MyTransfoXLConfig('small', tokenizer=tok).cutoffs == [1000]
    assert MyTransfoXLConfig('small', tokenizer=tok, cutoffs=[]).cutoffs == []      # kwargs override (train.py:526)
    assert MyTransfoXLConfig('small', tokenizer=MusicTokenizer(pitch_kind='midi')).cutoffs == []
    assert c.model_meta == dict(n_layer=12, hidden_size=768, ff_size=3072, seg_len=256, max_len=2048, vocab_size=c.vocab_size)


def test_param_layout_counts_and_alignment():
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
    from symbolic_music_generation_amd.xl_engine import ParamLayout
    lay = ParamLayout(MyTransfoXLConfig('base', vocab_size=418, cutoffs=[]))
    n = sum(math.prod(lay.entries[k][1]) for k in lay.real_names())
    assert n == 92_435_362                        # notebook/train/transformer-xl.ipynb:491 (92.4 M)
    for k, (off, shape) in lay.entries.items():
        if k not in ('crit.cluster_weight', 'crit.cluster_bias', '_pad.head_rows'):
            assert off % 8 == 0, k
    lay2 = ParamLayout(MyTransfoXLConfig('debug', vocab_size=1190, cutoffs=[1000]))
    e, cw = lay2.entries['transformer.word_emb.emb_layers.0.weight'], lay2.entries['crit.cluster_weight']
    assert cw[0] == e[0] + 1190 * 128 and lay2.head_rows_padded == 1192
    # weight-decay split: biases and LayerNorm weights live past n_decay
    for k, (off, _) in lay2.entries.items():
        nodecay = ('bias' in k) or ('layer_norm' in k)
        assert (off >= lay2.n_decay) == nodecay, k


def test_vocab_sizes_and_golden_ids():
    from symbolic_music_generation_amd.vocab import MusicTokenizer, MusicVocabulary
    assert [len(MusicVocabulary(pitch_kind=k)) for k in ('midi', 'step', 'degree')] == [422, 560, 1190]
    v = MusicVocabulary(pitch_kind='degree')
    assert [v.tok2id[t] for t in ('[OMIT]', '[PAD]', '<bar>', '</s>', '<melody>', '<bass>', '<tup>', '</tup>')] == list(range(8))
    assert v.toks['time_sig'] == ['TimeSig_rare', 'TimeSig_2/2', 'TimeSig_2/4', 'TimeSig_3/4', 'TimeSig_4/4', 'TimeSig_5/4',
                                  'TimeSig_6/8', 'TimeSig_12/8']
    assert len(v.toks['tempo']) == 203 and len(v.toks['key']) == 24 and len(v.toks['duration']) == 49
    assert v.toks['duration'][:4] == ['d_rare', 'd_1/8', 'd_1/4', 'd_3/8'] and v.toks['duration'][-1] == 'd_6'
    assert v.t2i('Tempo_20') == v.tok2id['Tempo_low'] and v.t2i('Tempo_300') == v.tok2id['Tempo_high']
    assert v.t2i('d_13') == v.tok2id['d_rare'] and v.t2i('TimeSig_7/8') == v.tok2id['TimeSig_rare']
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sample_score_ids.npz'))
    assert {k: len(g[k]) for k in g.files} == dict(sample_full_midi=2512, sample_full_step=4291, sample_full_degree=3226,
                                                   gen_broken=957)
    for name, kind in [('sample_full_midi', 'midi'), ('sample_full_step', 'step'), ('sample_full_degree', 'degree'),
                       ('gen_broken', 'degree')]:
        tok = MusicTokenizer(pitch_kind=kind)
        ids = g[name].astype(np.int64)
        assert ids.min() >= 0 and ids.max() < tok.vocab_size
        toks = tok.decode(ids).split()
        assert tok.encode(' '.join(toks)) == ids.tolist()                 # round trip
        # grammar: TimeSig Tempo [Key] ... pieces end with </s> except the broken generation (no EOS)
        assert toks[0].startswith('TimeSig_') and toks[1].startswith('Tempo_')
        assert (toks[-1] == '</s>') == (name != 'gen_broken')
    enc = MusicTokenizer(pitch_kind='midi')('TimeSig_4/4 Tempo_120 <bar> p_1/4 d_1', padding='max_length', truncation=True,
                                            max_length=8, return_tensors='pt')
    assert list(enc.keys()) == ['input_ids'] and enc['input_ids'].shape == (1, 8) and enc['input_ids'][0, -1].item() == 1


def test_schedule_and_trainargs():
    from symbolic_music_generation_amd.trainer import TrainArgs, lr_at, collate_clm, ntp_accuracy
    a = TrainArgs('transf-xl', 'base')()
    assert a['per_device_train_batch_size'] == 32 and a['learning_rate'] == 3e-4 and a['weight_decay'] == 1e-2
    assert a['lr_scheduler_type'] == 'cosine' and a['warmup_ratio'] == 0.1 and a['max_grad_norm'] == 1
    assert TrainArgs('transf-xl', 'debug')()['lr_scheduler_type'] == 'constant'
    ref = torch.optim.lr_scheduler.LambdaLR  # HF cosine-with-warmup closed form
    tot, base = 100, 3e-4
    for s in (0, 5, 10, 55, 99):
        warm = 10
        want = base * s / warm if s < warm else base * 0.5 * (1 + math.cos(math.pi * (s - warm) / (tot - warm)))
        assert abs(lr_at(s, tot, base, 'cosine', 0.1) - want) < 1e-12
    ids = torch.tensor([[5, 6, 1, 1]])
    _, lab = collate_clm(ids, pad_token_id=1)
    assert lab.tolist() == [[5, 6, -100, -100]]
    assert ntp_accuracy(torch.tensor([[6, 9, 9, 9]]), lab) == 1.0


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from symbolic_music_generation_amd.dist import GradSync, layer_buckets
from symbolic_music_generation_amd.transformer_xl import MyTransfoXLConfig
from symbolic_music_generation_amd.xl_engine import ParamLayout
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:' + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
cfg = MyTransfoXLConfig('debug', vocab_size=1190, cutoffs=[1000], n_layer=3)
class E: pass
e = E(); e.cfg = cfg; e.layout = ParamLayout(cfg)
r = dist.get_rank()
e.G = torch.arange(e.layout.total, dtype=torch.float32) * (r + 1)
gs = GradSync(e)
per_layer, rest = layer_buckets(e.layout, cfg.n_layer)
cover = torch.zeros(e.layout.total)
for sl in per_layer:
    for lo, hi in sl: cover[lo:hi] += 1
for lo, hi in rest: cover[lo:hi] += 1
assert (cover == 1).all(), 'buckets must tile the flat buffer exactly once'
for l in reversed(range(cfg.n_layer)): gs.layer_done(l)
gs.finish()
want = torch.arange(e.layout.total, dtype=torch.float32) * 3
assert torch.equal(e.G, want), 'all-reduce(sum) over 2 ranks'
dist.destroy_process_group()
print('ok')
'''


def test_data_parallel_gradsync_gloo_world2(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_DP_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all('ok' in o for o in outs)


def test_train_and_my_train_args_surface():
    """TrainArgs.__call__(train_args, my_train_args, train_dataset) of the reference (train.py:192-246)"""
    from symbolic_music_generation_amd.trainer import get_train_and_my_train_args
    ds = list(range(1000))
    args, my = get_train_and_my_train_args('transf-xl', 'base', dict(per_device_train_batch_size=21, save_strategy='epoch'),
                                           dict(logging_strategy='epoch', save_epochs=4, random_crop=32), ds)
    assert my['steps_per_epoch'] == math.ceil(1000 / 21) == 48 and my['logging_steps'] == 48
    assert args['save_strategy'] == 'steps' and args['save_steps'] == 4 * 48            # save every 4 epochs (:214-221)
    assert my['insert_key'] is False and my['proportional_mixing'] is False and my['random_crop'] == 32 and my['tqdm'] is False
    assert args['per_device_train_batch_size'] == 21 and args['learning_rate'] == 3e-4 and args['max_grad_norm'] == 1
    assert all(v is not None for v in args.values())
    with pytest.raises(ValueError):
        get_train_and_my_train_args('reformer', 'tiny', None, dict(logging_strategy='sometimes'), ds)


def test_shape_predicates_of_the_round3_paths():
    """host-side decisions that pick a kernel path (no GPU): where the forward's phantom value-sum applies, and for which output
    sizes the large-tile GEMM offers its relu-mask bit buffer (M * N / 8 bytes; 0 = the engines fall back to the activations)"""
    from symbolic_music_generation_amd import ops
    ok = ops.phantom_sum_applies
    assert ok(T=2048, dh=64, M=2048, Kc=2048)                  # C3, zero memories
    assert ok(T=768, dh=64, M=1024, Kc=768 + 192)              # partial memories
    assert not ok(T=2048, dh=64, M=2048, Kc=4096)              # full real memories: no phantom distance
    assert not ok(T=2048, dh=32, M=2048, Kc=2048) and not ok(T=2000, dh=64, M=2048, Kc=2000) and not ok(T=512, dh=64, M=384, Kc=512)
    nbytes = ops.gemm_relu_mask_bytes
    assert nbytes(131072, 3072) == 131072 * 3072 // 8           # C3 FFN
    assert nbytes(131072, 2048) == 131072 * 2048 // 8           # C4 FFN
    assert nbytes(1000, 3072) == 0 and nbytes(2048, 200) == 0   # ragged
    assert nbytes(2048, 3072) == 0                              # fewer tiles than one round of the chip: 192-wide tiles win


def test_recorded_pmc_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` comes from a committed PMC record (counters cannot be collected inside the timed run): the
    record carries a hash of the translation units that hold the measured kernels, and bench.py reports the bytes only while that
    hash matches the sources -- otherwise `traffic` is null and the source note says the record is stale."""
    import json
    import shutil
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import bench
    import pmc_traffic
    sha = pmc_traffic.sources_sha16('train')
    assert len(sha) == 16 and sha == pmc_traffic.sources_sha16('train')
    prof = tmp_path / 'profiles'
    prof.mkdir()
    names = ['fused_delta_kernel', 'relattn_bwd_fused_kernel<1>', 'relattn_dq_finish_kernel', 'relattn_drd_phantom_kernel']
    rec = {'per_gpu_batch': 64, 'group_sources_sha16': sha, 'kernels': {n: {'hbm_bytes_per_launch': 1.0e9} for n in names}}
    (prof / 'r99_c3_pmc_traffic.json').write_text(json.dumps(rec))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    # (the hash is taken from the real sources: bench imports pmc_traffic from ROOT/scripts, so give the fake root a copy)
    shutil.copytree(os.path.join(ROOT, 'scripts'), tmp_path / 'scripts', ignore=shutil.ignore_patterns('__pycache__', 'ubench'))
    got, src = bench.pmc_traffic('c3', 64)
    assert got == 4.0e9 and src == 'r99_c3_pmc_traffic.json'
    assert bench.pmc_traffic('c3', 32) == (None, None)           # another batch: not this measurement
    rec['group_sources_sha16'] = '0' * 16
    (prof / 'r99_c3_pmc_traffic.json').write_text(json.dumps(rec))
    got, src = bench.pmc_traffic('c3', 64)
    assert got is None and 'STALE' in src


def test_no_backward_path_regenerates_the_ffn_hidden_dropout_mask():
    """MXL_GEMM_RELU | MXL_GEMM_DROPOUT (the FFN's hidden activations) draws a mask of its own -- one hash per element pair, 16-bit
    decisions (include/musicxl.h, MXL_GEMM_DROPOUT) -- not mxl_dropout_bf16's keep-mask.  That is only sound while nothing regenerates
    the mask of that site: the backward must take it from the saved bits or from the zeros of the stored activations.  Pin it: in the
    engines the hidden-activation site (_site(l, 1) / the Reformer's ffn site) appears in exactly one call, the forward GEMM."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parents[1] / 'symbolic_music_generation_amd'
    xl = (root / 'xl_engine.py').read_text()
    uses = [m.start() for m in re.finditer(r'self\._site\(l, 1\)', xl)]
    assert len(uses) == 1, 'the FFN-hidden dropout site must be used by the forward GEMM only'
    call = xl[xl.rfind('ops.', 0, uses[0]):uses[0]]
    assert call.startswith('ops.gemm('), call[:40]
    # the regenerating kernels (stand-alone dropout, LayerNorm backward, embedding backward) never name that site
    for m in re.finditer(r'ops\.(dropout|ln_residual_bwd\w*|embed_bwd)\(', xl):
        stmt = xl[m.start():xl.find('\n\n', m.start())]
        assert '_site(l, 1)' not in stmt.split(')\n')[0]



def test_load_trained_key_table_and_argument_checks(tmp_path):
    """trainer.load_trained mirrors musicnlp.trainer.eval.load_trained (eval.py:32-95): the author's run table resolves a model_key
    to <base>/models/<run>/<checkpoint>, a model_key also names the model, and the argument errors come before any GPU work"""
    from symbolic_music_generation_amd import trainer
    tab = trainer.TRAINED_KEY2PATH['full']
    assert len(tab) == 18
    assert tab[('reformer', 'P&M', '256-256ep', 'mid-pch')] == ['2022-10-03_11-58-11_reformer', 'trained']
    assert tab[('transf-xl', 'All', '128ep', 'midi_longer-seq')] == ['2022-11-21_21-22-24_transf-xl', 'checkpoint-30348']
    assert sum(k[0] == 'reformer' for k in tab) == 4 and sum(k[0] == 'transf-xl' for k in tab) == 14
    with pytest.raises(NotImplementedError):
        trainer.load_trained('transf-xl', 'x', mode='melody')
    with pytest.raises(KeyError):
        trainer.load_trained(model_key=('transf-xl', 'All', '1ep', 'nope'), base_path=str(tmp_path))
    with pytest.raises(ValueError):
        trainer.load_trained('gpt2', 'x', base_path=str(tmp_path))
    with pytest.raises(ValueError):
        trainer.load_trained('transf-xl', base_path=str(tmp_path))
    # a key resolves to the run directory below <base>/models (nothing is there: the config read fails with that path in the message)
    with pytest.raises(Exception) as e:
        trainer.load_trained(model_key=('transf-xl', 'All', '128ep', 'midi'), base_path=str(tmp_path))
    assert os.path.join(str(tmp_path), 'models', '2022-11-14_13-04-30_transf-xl', 'trained') in str(e.value)


def test_gemm_w4_register_table_is_what_its_generator_writes():
    """csrc/gemm_w4_gen.inc (the four-wave NT GEMM's instruction streams with every accumulator pinned to a[4 (8 i + j) : + 3]) is
    generated: the committed file must be scripts/gen_gemm_w4.py's output, and every accumulator register must appear exactly once
    per MFMA stream"""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'gen_gemm_w4.py'), '--check'])
    assert r.returncode == 0, 'run python scripts/gen_gemm_w4.py and commit csrc/gemm_w4_gen.inc'
    text = open(os.path.join(root, 'symbolic_music_generation_amd', 'csrc', 'gemm_w4_gen.inc')).read()
    regs = [(int(i), int(j), int(lo), int(hi)) for i, j, lo, hi in re.findall(r'W4_MFMA\((\d+), (\d+), (\d+), (\d+)\)', text)]
    assert len(regs) == 64 and sorted(lo for _, _, lo, _ in regs) == list(range(0, 256, 4))
    assert all(lo == 4 * (8 * i + j) and hi == lo + 3 for i, j, lo, hi in regs)
    assert len(re.findall(r'W4_PIECE\(\d\)', text)) == 8 and len(re.findall(r'w4_rd<\d+>', text)) == 16
