"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py ...
    python scripts/pmc_mfma.py gpurun_out/pmc_mfma/m_counter_collection.csv gpurun_out/pmc_mfma/m_kernel_trace.csv profiles/r01_c3_mfma_util.json

MI355X_MICROARCH.md (rocprofv3 PMC slots / cycle constants): ROCm 7.2 has no gfx950 derived-metric section, so the ratio is
formed here.  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over SIMDs (32 per
v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16: the dense bf16 rate); GRBM_GUI_ACTIVE is reported as the sum over the
8 XCDs, so elapsed shader cycles = GRBM_GUI_ACTIVE / 8 and
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).
`mfma_util` is the share of the chip's matrix-pipe cycles a kernel occupies at the clock it actually ran at (the clock
under the profiler is lower than unprofiled, so compare ratios, not times); `eff_clock_ghz` = cycles / kernel duration."""
import json
import sys

import pandas as pd

CUS, SIMDS, XCDS = 256, 4, 8


def short(s):
    return (s.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True)
            .str.replace('void ', ''))


def main(counter_csv, trace_csv, out_json, batch=32):
    df = pd.read_csv(counter_csv)
    df = df.assign(k=short(df.Kernel_Name))
    piv = df.pivot_table(index=['Dispatch_Id', 'k'], columns='Counter_Name', values='Counter_Value', aggfunc='sum').reset_index()
    tr = pd.read_csv(trace_csv)
    tr = tr.assign(dur=tr.End_Timestamp - tr.Start_Timestamp)[['Dispatch_Id', 'dur']]
    piv = piv.merge(tr, on='Dispatch_Id', how='left')
    out = {}
    tot_busy = tot_cyc = 0.0
    for k, g in piv.groupby('k'):
        busy = float(g.SQ_VALU_MFMA_BUSY_CYCLES.sum())
        cyc = float(g.GRBM_GUI_ACTIVE.sum()) / XCDS
        tot_busy += busy
        tot_cyc += cyc
        if cyc <= 0:
            continue
        out[k] = {'launches': int(len(g)), 'mfma_busy_cycles_per_launch': busy / len(g), 'shader_cycles_per_launch': cyc / len(g),
                  'avg_duration_us': float(g.dur.mean()) / 1e3, 'eff_clock_ghz': cyc / max(float(g.dur.sum()), 1.0),
                  'mfma_util': busy / (cyc * CUS * SIMDS)}
    json.dump({'command': 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (SURVEY C3, 1x MI355X)',
               'per_gpu_batch': int(batch), 'formula': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)',
               'whole_run_mfma_util': tot_busy / (tot_cyc * CUS * SIMDS), 'kernels': out}, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
