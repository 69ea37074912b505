"""Build libmusicxl.so (every HIP translation unit under csrc/) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on the MI355X box.
The resulting .so is git-ignored but travels with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libmusicxl.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-ffp-contract=fast']
# VGPR-form MFMAs everywhere.  By default hipcc gives every MFMA of a kernel that exceeds 256 registers an AGPR destination; the
# attention-backward query-owner kernel (one workgroup per CU, > 256 registers) then pays ~220 v_accvgpr_write/read per tile to
# zero the S / dP / G accumulators and to bring the results back for the VALU work.  With VGPR-form MFMAs the AGPRs only hold the
# few values that do not fit (593 -> 204 such moves in that kernel, 338 -> 303 registers, no scratch); the other kernels are
# unchanged or slightly better.
FLAGS += ['-mllvm', '-amdgpu-mfma-vgpr-form']
# AMDGPU-specific register-pressure trackers in the machine scheduler: A/B on the attention backward 2.27 -> 2.17 ms per layer,
# whole C3 step +1.9 % (max-ilp / max-memory-clause strategies and the occupancy bias measured neutral or worse).
FLAGS += ['-mllvm', '-amdgpu-use-amdgpu-trackers']
# per translation unit.  relattn_fwd.hip: the max-ilp strategy of the machine scheduler, -1 % on the forward kernel over six alternating
# same-box runs (1.605 -> 1.589 ms; iterative-ilp -0.4 %, max-memory-clause +0.2 %); on the fused backward the same flag is within noise
# (profiles/r06_fused_bwd_setprio_ab.log, r06_fwd_setprio_ab.log)
# reformer.hip: the same flag, chunk-attention forward 172 -> 169 us, query-owner backward 264 -> 243 us, key-owner 360 -> 357 us per
# launch at C4, the leg +0.5 % (4.850 / 4.817 -> 4.869 / 4.850 M tok/s alternating; profiles/r06_rf_ab2_max_ilp.log).  Not for the
# phantom dRd kernel (0.659 -> 0.664 ms).
EXTRA_FLAGS = {'relattn_fwd.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'], 'reformer.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h') or f.endswith('.inc')]
    hdrs.append(os.path.join(HERE, '..', 'include', 'musicxl.h'))
    objs, jobs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(s, []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
