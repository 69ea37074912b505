"""Per-kernel time of two rocprofv3 kernel traces side by side (sum over the run, ms): python scripts/trace_diff.py a.csv b.csv"""
import sys

import pandas as pd


def load(path):
    df = pd.read_csv(path)
    name = df.Kernel_Name.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    df = df.assign(k=name.str[:60], dur=(df.End_Timestamp - df.Start_Timestamp) / 1e6)
    return df.groupby('k').dur.agg(['count', 'sum'])


a, b = load(sys.argv[1]), load(sys.argv[2])
t = a.join(b, lsuffix='_a', rsuffix='_b', how='outer').fillna(0)
t['delta_ms'] = t.sum_b - t.sum_a
t['ratio'] = t.sum_b / t.sum_a.where(t.sum_a > 0)
pd.set_option('display.width', 200, 'display.max_rows', 200, 'display.float_format', lambda v: f'{v:.3f}')
print(t.sort_values('delta_ms', ascending=False).head(25).to_string())
print('total a', round(a['sum'].sum(), 2), 'ms   total b', round(b['sum'].sum(), 2), 'ms')
