"""Aggregate the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs -- both do not fit the TCC slots) of
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` into per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
        profiles/r01_c3_pmc_traffic.json [per-GPU batch of that run, default 32]

Corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide (16 B/lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact for
16-byte stores and float atomics.  Infinity-Cache hits are included in both (memory-side L2 counters)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the translation units that hold the kernels of bench.py's roofline group: their hash goes into the record, and bench.py reports
# the recorded traffic only while it still matches the sources (a changed kernel must be re-measured)
GROUP_SOURCES = {'train': ['relattn_bwd_fused.hip', 'relattn_drd_phantom.hip', 'relattn_fwd.hip'], 'reformer': ['gemm.hip'], 'decode': ['decode.hip']}


def sources_sha16(kind):
    h = hashlib.sha256()
    for f in GROUP_SOURCES[kind]:
        h.update(open(os.path.join(ROOT, 'symbolic_music_generation_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    import pandas as pd
    df = pd.read_csv(path)
    df = df[df.Counter_Name == counter]
    df = df.assign(k=df.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True)
                   .str.replace('void ', ''))
    return df.groupby('k').Counter_Value.agg(['count', 'mean'])


def main(fetch_csv, write_csv, out_json, batch=32, kind='train', prompt_len=256):
    prompt_len = int(prompt_len)
    f = per_kernel(fetch_csv, 'FETCH_SIZE')
    w = per_kernel(write_csv, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(f.index) | set(w.index)):
        fk = float(f['mean'].get(k, 0.0)) * 1000.0
        wk = float(w['mean'].get(k, 0.0)) * 1000.0
        out[k] = {'launches': int(f['count'].get(k, w['count'].get(k, 0))), 'fetch_size_bytes_raw': fk,
                  'fetch_bytes_corrected_x2': 2.0 * fk, 'write_bytes': wk, 'hbm_bytes_per_launch': 2.0 * fk + wk}
    rec = {'command': 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (SURVEY C3, 1x MI355X)',
           'per_gpu_batch': int(batch), 'units': 'bytes per launch (counter mean over launches x 1000)',
           'group_sources': GROUP_SOURCES[kind], 'group_sources_sha16': sources_sha16(kind), 'kernels': out}
    if kind == 'decode':
        # eager decode window (bench.py --mode decode --eager --decode-steps 40): bytes of the kernels of the decode loop per step
        loop = ['advance_kernel', 'decode_attn_kernel<64>', 'decode_bd_kernel', 'decode_embed_kernel', 'gemm_skinny_kernel',
                'ln_res_fwd_kernel<2>', 'ln_res_partial_fwd_kernel<2>', 'logprob_full_kernel', 'sample_kernel', 'sample_step_kernel']
        # one launch per step (round 6: the sampler launch also advances the counters)
        steps = float(out['advance_kernel' if 'advance_kernel' in out else 'sample_step_kernel']['launches'])
        d, L, V, B = 768, 12, 1190, int(batch)
        first = prompt_len + 1 + 3   # prompt, its sample, warm-up steps (bench.decode_leg): first ring slot count of the window
        valid = sum(min(first + i, 2048) for i in range(int(steps))) / steps
        rec.update(command=f'python3 bench.py --mode decode --eager --decode-steps 40 --decode-prompt {prompt_len} --no-cpu-baseline (SURVEY C5, eager launches: '
                           'rocprofv3 cannot collect counters over hipGraph replays)',
                   decode_loop_kernels=loop, decode_steps_profiled=steps,
                   positions=f'prompt {prompt_len} + warm-up: ring slots ~{first}..{first + int(steps)} written',
                   hbm_bytes_per_decode_step=sum(out[k]['hbm_bytes_per_launch'] * out[k]['launches'] for k in loop if k in out) / steps,
                   algorithmic_bytes_per_step_at_these_positions=L * 12 * d * d * 2 + V * d * 2 + B * L * 2 * valid * d * 2)
    elif kind == 'reformer':
        rec['command'] = 'python3 bench.py --mode reformer --no-cpu-baseline --steps 2 --warmup 1 (SURVEY C4, 1x MI355X)'
    json.dump(rec, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:7])
