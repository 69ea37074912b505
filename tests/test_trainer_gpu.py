"""End-to-end: factory -> MyTrainer epochs (train + eval + checkpoint) -> save/load round trip -> generate, for both model
families, through the drop-in surface (musicnlp/trainer/train.py:31-59, 287-368 call pattern)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy_dataset(tok, n, T, seed=0):
    """grammar-shaped synthetic pieces: TimeSig Tempo (<bar> <melody> (pitch dur)*)* </s>, padded to T"""
    g = torch.Generator().manual_seed(seed)
    v = tok.vocab
    pitches = [v.tok2id[t] for t in v.toks['pitch'][2:60]]
    durs = [v.tok2id[t] for t in v.toks['duration'][1:9]]
    out = []
    for _ in range(n):
        ids = [v.tok2id['TimeSig_4/4'], v.tok2id['Tempo_120']]
        while len(ids) < T - 12:
            ids += [v.tok2id['<bar>'], v.tok2id['<melody>']]
            for _ in range(4):
                ids += [pitches[torch.randint(0, 8, (1,), generator=g).item()], durs[torch.randint(0, 2, (1,), generator=g).item()]]
        ids.append(tok.eos_token_id)
        ids += [tok.pad_token_id] * (T - len(ids))
        out.append(torch.tensor(ids[:T]))
    return out


def test_transfxl_train_eval_save_generate(dev, tmp_path):
    from symbolic_music_generation_amd.trainer import get_model_n_tokenizer, MyTrainer
    from symbolic_music_generation_amd.transformer_xl import MyTransfoXLLMHeadModel
    tok, model, meta = get_model_n_tokenizer('transf-xl', 'debug', pitch_kind='midi',
                                             model_config=dict(max_length=128, mem_len=64, n_layer=2, cutoffs=[], dropout=0.0),
                                             device=dev)
    assert meta['model name'] == 'TransformerXl' and meta['max length'] == 128 and tok.model_max_length == 128
    train, evals = _toy_dataset(tok, 32, 128), _toy_dataset(tok, 8, 128, seed=1)
    logs = []
    tr = MyTrainer(model, tok, train, evals, model_name='transf-xl', model_size='debug', log_fn=logs.append,
                   train_args=dict(per_device_train_batch_size=8, per_device_eval_batch_size=8, num_train_epochs=6,
                                   learning_rate=3e-3, output_dir=str(tmp_path), load_best_model_at_end=False))
    res = tr.train()
    ev = [d for d in logs if 'eval_loss' in d]
    assert res['global_step'] == 24 and len(ev) == 6
    assert ev[-1]['eval_loss'] < 0.7 * ev[0]['eval_loss'] and ev[-1]['eval_ntp_acc'] > ev[0]['eval_ntp_acc']
    ck = os.path.join(str(tmp_path), 'checkpoint-24')
    assert os.path.exists(os.path.join(ck, 'pytorch_model.bin')) and os.path.exists(os.path.join(ck, 'config.json'))
    m2 = MyTransfoXLLMHeadModel.from_pretrained(ck, device=dev).eval()
    ids = evals[0][None].to(dev)
    a = model.eval()(input_ids=ids).logits
    b = m2(input_ids=ids).logits
    assert torch.equal(a, b)
    prompt = ids[:, :10]
    g1 = model.generate(input_ids=prompt, max_length=40, do_sample=False)
    g2 = m2.generate(input_ids=prompt, max_length=40, do_sample=False)
    assert torch.equal(g1, g2) and g1.shape == (1, 40) and torch.equal(g1[:, :10], prompt)
    s = model.generate(input_ids=prompt.repeat(4, 1), max_length=40, do_sample=True, top_k=8, temperature=1.0)
    assert s.shape == (4, 40) and (s >= 0).all() and (s < tok.vocab_size).all()


def test_reformer_train_eval(dev, tmp_path):
    from symbolic_music_generation_amd.trainer import get_model_n_tokenizer, MyTrainer
    tok, model, meta = get_model_n_tokenizer('reformer', 'debug-large', pitch_kind='midi',
                                             model_config=dict(max_position_embeddings=256, axial_pos_shape=(16, 16),
                                                               attn_layers=['local', 'lsh'] * 2, num_hashes=2), device=dev)
    assert meta['model name'] == 'Reformer' and meta['max length'] == 256 and meta['attention_shape'] == '8x16'
    assert model.config.eos_token_id == tok.eos_token_id and model.config.pad_token_id == tok.pad_token_id
    train, evals = _toy_dataset(tok, 16, 256), _toy_dataset(tok, 8, 256, seed=1)
    logs = []
    tr = MyTrainer(model, tok, train, evals, model_name='reformer', model_size='debug-large', log_fn=logs.append,
                   train_args=dict(per_device_train_batch_size=8, per_device_eval_batch_size=8, num_train_epochs=10,
                                   learning_rate=3e-3, load_best_model_at_end=False))
    tr.train()
    ev = [d for d in logs if 'eval_loss' in d]
    assert ev[-1]['eval_loss'] < 0.85 * ev[0]['eval_loss']


@pytest.mark.gpu
def test_bench_rccl_path_single_rank():
    """bench.py under torch.distributed.run with the collective path forced on one rank: RCCL init, the per-layer async
    gradient all-reduces overlapped with the backward, barrier + MAX-over-ranks timing, one JSON line."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MXL_DIST_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--workload', 'tiny', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['unit'] == 'tokens/s'
