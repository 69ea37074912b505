"""Where a tile of the query-owner attention backward spends its cycles: runs scripts/perf_attn_layer.py's layer against the
diagnostic build (scripts/ab_build.sh relattn_bwd stamp -DMXL_STAMP; MXL_LIB_PATH=.../build/libmusicxl_stamp.so) and prints the
per-segment shares of wave cycles summed over all waves."""
import ctypes as C, os, sys, runpy
import torch  # noqa: F401  (first: libmusicxl must bind to the HIP runtime torch has loaded, not to a second copy)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symbolic_music_generation_amd import _lib
L = _lib.lib()
out = (C.c_ulonglong * 16)()
raw = C.CDLL(_lib.LIB_PATH)
os.environ['ITERS'] = '3'
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'perf_attn_layer.py'))      # warm-up (first touch of the buffers)
assert raw.mxl_debug_dq8_stamps(out) == 0                 # read and zero
os.environ['ITERS'] = '2'
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'perf_attn_layer.py'))
rc = raw.mxl_debug_dq8_stamps(out)
assert rc == 0, rc
names = ['0 tile top: loads issued', '1 S / dP MFMA chains', '2 skew read (BD) + wait', '3 exp, mul, pack, skew write (dS)', '4 dQw MFMAs (K^T reads)',
         '5 store next K/V/Rd to LDS (vmcnt)', '6 barrier 1', '7 phase 3: dG stores + dQr MFMAs', '8 pre-phase: G blocks of next tile',
         '9 barrier 2', '10 phantom-distance loop', '11 prologue']
v = [out[i] for i in range(12)]
tot = float(sum(v))
for n, x in zip(names, v):
    print(f'{n:45s} {x:16d}  {100.0 * x / tot:5.1f} %')
