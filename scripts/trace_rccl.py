"""Where RCCL's kernels land in a rocprofv3 kernel trace of the C3 training step (scripts/r05_dist_overlap.sh): every kernel whose name
mentions nccl / rccl, its duration, its queue, and the compute kernels whose [start, end) interval overlaps it."""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
name = df.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
df = df.assign(k=name, dur=(df.End_Timestamp - df.Start_Timestamp) / 1e3)
is_cc = df.k.str.contains('nccl|rccl', case=False)
cc = df[is_cc]
print(f'{len(df)} kernel launches, {len(cc)} of them RCCL kernels; queues: {sorted(df.Queue_Id.unique().tolist()) if "Queue_Id" in df else "?"}')
if not len(cc):
    print('no RCCL kernel in the trace: at world size 1 the in-place all-reduce is a no-op on the device (nothing to overlap, nothing to measure)')
    sys.exit(0)
print(cc.groupby('k').dur.agg(['count', 'mean', 'sum']).to_string())
other = df[~is_cc]
tot_overlap = 0.0
for _, r in cc.head(12).iterrows():
    ov = other[(other.Start_Timestamp < r.End_Timestamp) & (other.End_Timestamp > r.Start_Timestamp)]
    t = (ov[['End_Timestamp']].clip(upper=r.End_Timestamp).End_Timestamp - ov[['Start_Timestamp']].clip(lower=r.Start_Timestamp).Start_Timestamp).sum() / 1e3
    tot_overlap += t
    print(f'  {r.k[:50]:50s} {r.dur:9.1f} us  beside: {", ".join(sorted(set(ov.k.str[:28]))[:4]) or "nothing"}')
