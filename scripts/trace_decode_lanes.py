"""Round 6: where a decode lane's dependent chain spends its time.  From a rocprofv3 kernel trace of the two-lane decode leg

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dtrace -o d -- python3 bench.py --mode decode --no-cpu-baseline --decode-steps 300
    python scripts/trace_decode_lanes.py gpurun_out/dtrace/.../d_kernel_trace.csv

kernels are attributed to a lane by their queue / stream id; per lane and kernel kind: mean duration, mean wait since the previous
kernel of the SAME lane ended (= node-to-node latency of the chain, launch + scheduling behind the other lane's work), and how much of
its duration overlaps a ring-attention kernel of the OTHER lane."""
import sys
import numpy as np
import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
name = df.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
df = df.assign(k=name, dur=(df.End_Timestamp - df.Start_Timestamp) / 1e3)
qcol = 'Queue_Id' if 'Queue_Id' in df.columns else ('Stream_Id' if 'Stream_Id' in df.columns else None)
print('columns:', list(df.columns))
att = df[df.k.str.startswith('decode_attn_kernel')]
# the timed region: the last 60 % of the attention launches (prompt pass and capture excluded)
t0 = att.Start_Timestamp.quantile(0.4)
df = df[df.Start_Timestamp >= t0]
lanes = df.groupby(qcol).size().sort_values(ascending=False).index[:2].tolist() if qcol else [None]
print('lanes (by', qcol, '):', lanes)
att_by_lane = {q: df[(df[qcol] == q) & df.k.str.startswith('decode_attn_kernel')][['Start_Timestamp', 'End_Timestamp']].values for q in lanes}
for q in lanes:
    w = df[df[qcol] == q].reset_index(drop=True)
    wait = np.r_[0.0, (w.Start_Timestamp.values[1:] - w.End_Timestamp.values[:-1]) / 1e3]
    other = att_by_lane[[x for x in lanes if x != q][0]] if len(lanes) > 1 else np.zeros((0, 2))
    ov = np.zeros(len(w))
    if len(other):
        starts, ends = other[:, 0], other[:, 1]
        for i, (s, e) in enumerate(zip(w.Start_Timestamp.values, w.End_Timestamp.values)):
            j0 = np.searchsorted(ends, s)
            j = j0
            tot = 0.0
            while j < len(starts) and starts[j] < e:
                tot += max(0.0, min(e, ends[j]) - max(s, starts[j]))
                j += 1
            ov[i] = tot / max(e - s, 1)
    w = w.assign(wait=wait, ov=ov)
    span = (w.End_Timestamp.max() - w.Start_Timestamp.min()) / 1e3
    nstep = (w.k.str.startswith('sample')).sum()
    print(f'lane {q}: {len(w)} launches, {nstep} steps, {span / max(nstep, 1):.1f} us per step; kernel time {w.dur.sum() / max(nstep, 1):.1f} us, '
          f'waits {w.wait.clip(upper=200).sum() / max(nstep, 1):.1f} us per step')
    g = w.groupby('k').agg(n=('dur', 'size'), dur=('dur', 'mean'), wait=('wait', lambda x: x.clip(upper=200).mean()), ov=('ov', 'mean'))
    g = g.assign(per_step_us=(g.dur + g.wait) * g.n / max(nstep, 1)).sort_values('per_step_us', ascending=False)
    print(g.head(12).round(2).to_string())
