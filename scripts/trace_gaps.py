"""GPU idle time inside the C3 training step from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -o g -- python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2
    python scripts/trace_gaps.py gpurun_out/gaps/.../g_kernel_trace.csv

Steps are delimited by the AdamW launches; prints, per step, wall time, the sum of kernel durations and the largest gaps."""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
name = df.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
df = df.assign(k=name, dur=df.End_Timestamp - df.Start_Timestamp)
ad = df.index[df.k.str.startswith('adamw_kernel')].tolist()
print('adamw launches:', len(ad))
for a, b in zip(ad[:-1], ad[1:]):
    w = df.iloc[a + 1:b + 1]
    wall = (w.End_Timestamp.max() - df.End_Timestamp[a]) / 1e6
    busy = w.dur.sum() / 1e6
    gaps = (w.Start_Timestamp.values[1:] - w.End_Timestamp.values[:-1]) / 1e3
    big = sorted(((g, w.k.values[i], w.k.values[i + 1]) for i, g in enumerate(gaps) if g > 20), reverse=True)[:6]
    print(f'step: wall {wall:.2f} ms, kernels {busy:.2f} ms ({len(w)} launches), idle {wall - busy:.2f} ms; gaps > 20 us: {sum(1 for g in gaps if g > 20)}, '
          f'mean gap {gaps.mean():.1f} us')
    for g, x, y in big:
        print(f'    {g:8.1f} us between {x[:40]} and {y[:40]}')
