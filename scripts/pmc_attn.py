"""Issue / wait anatomy of the attention kernels at one C3 layer (B = 16): two rocprofv3 PMC passes of `perf_attn_layer.py`

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
              --kernel-trace --output-format csv -d gpurun_out/pa1 -o p -- python3 scripts/perf_attn_layer.py
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 \
              --kernel-trace --output-format csv -d gpurun_out/pa2 -o p -- python3 scripts/perf_attn_layer.py
    python scripts/pmc_attn.py gpurun_out/pa1/p_counter_collection.csv gpurun_out/pa2/p_counter_collection.csv

prints, per kernel, the counters per launch and the derived shares (SQ_WAVE_CYCLES etc. count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs: MI355X_MICROARCH.md cycle constants)."""
import sys

import pandas as pd


def short(s):
    return (s.str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True)
            .str.replace('void ', ''))


def main(*csvs):
    frames = []
    for c in csvs:
        df = pd.read_csv(c)
        df = df.assign(k=short(df.Kernel_Name))
        frames.append(df.pivot_table(index='k', columns='Counter_Name', values='Counter_Value', aggfunc='mean'))
    t = pd.concat(frames, axis=1)
    t = t[[c for c in t.columns]]
    pd.set_option('display.width', 250, 'display.max_columns', 40, 'display.float_format', lambda v: f'{v:,.0f}')
    keep = [k for k in t.index if 'relattn' in k or 'chunk_attn' in k or 'gemm' in k]
    print(t.loc[keep].T)
    if 'SQ_WAVE_CYCLES' in t:
        d = t.loc[keep]
        out = pd.DataFrame({
            'wait_any/wave': d.SQ_WAIT_ANY / d.SQ_WAVE_CYCLES,
            'wait_inst/wave': d.SQ_WAIT_INST_ANY / d.SQ_WAVE_CYCLES,
            'active_inst/wave': d.SQ_ACTIVE_INST_ANY / d.SQ_WAVE_CYCLES,
            'valu_active/wave': d.SQ_ACTIVE_INST_VALU / d.SQ_WAVE_CYCLES,
            'lds_active/wave': d.SQ_ACTIVE_INST_LDS / d.SQ_WAVE_CYCLES,
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 CUs = 32 SIMDs each); SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs
            # (= MFMA instructions x their cycles).  Rounds 2-4 divided by 16 here and reported twice the occupancy.
            'mfma_busy/simd_cycles': d.SQ_VALU_MFMA_BUSY_CYCLES / (d.SQ_BUSY_CYCLES * 32) if 'SQ_BUSY_CYCLES' in d else None,
        })
        pd.set_option('display.float_format', lambda v: f'{v:.3f}')
        print(out)
    if 'SQ_INSTS_MFMA' in t:
        d = t.loc[keep]
        pd.set_option('display.float_format', lambda v: f'{v:.2f}')
        print(pd.DataFrame({'valu/mfma': d.SQ_INSTS_VALU / d.SQ_INSTS_MFMA, 'lds/mfma': d.SQ_INSTS_LDS / d.SQ_INSTS_MFMA,
                            'trans/mfma': d.SQ_INSTS_VALU_TRANS_F32 / d.SQ_INSTS_MFMA, 'salu/mfma': d.SQ_INSTS_SALU / d.SQ_INSTS_MFMA,
                            'lds_conflict/idx_active': d.SQ_LDS_BANK_CONFLICT / d.SQ_LDS_IDX_ACTIVE}))


if __name__ == '__main__':
    main(*sys.argv[1:])
