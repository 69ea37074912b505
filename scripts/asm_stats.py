"""Static instruction mix of the kernels in one csrc/*.hip translation unit (compiled with the library's flags, -save-temps).

    python scripts/asm_stats.py relattn_bwd.hip dq8          # kernels whose mangled name contains 'dq8'

Prints per kernel: instruction counts by class (MFMA / VALU / LDS / SALU / memory), registers, scratch operations.  Static counts
say nothing about trip counts; they are for A/B-ing two versions of the same loop structure."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from symbolic_music_generation_amd import build as B
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    out = tempfile.mkdtemp(prefix='asm_stats_')
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(B.CSRC, src), '-o', os.path.join(out, 'x.o'),
                                                              '-save-temps']
    r = subprocess.run(cmd, cwd=out, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr)
        sys.exit(1)
    asm = [f for f in os.listdir(out) if f.endswith('gfx950.s')][0]
    s = open(os.path.join(out, asm)).read()
    print('asm:', os.path.join(out, asm))
    for name in re.findall(r'^(_Z\S+):', s, re.M):
        if pat not in name or '.end_amdhsa_kernel' not in s.split(name + ':', 1)[1]:
            continue
        body = s.split(name + ':', 1)[1].split('.end_amdhsa_kernel', 1)[0]
        code = body.split('.section', 1)[0]
        c = Counter()
        for line in code.splitlines():
            t = line.strip()
            if not line.startswith('\t') or not t or t[0] in '.;':
                continue
            i = t.split()[0]
            k = ('mfma' if 'mfma' in i else 'lds' if i.startswith('ds_') else 'valu' if i.startswith('v_') else
                 'salu' if i.startswith('s_') else 'mem' if i.startswith(('global', 'buffer', 'scratch', 'flat')) else 'other')
            c[k] += 1
            if i.startswith('scratch_'):
                c['scratch'] += 1
        regs = {k: (re.search(r'\.amdhsa_' + k + r'\s+(\d+)', body) or [None, '?'])[1] for k in ('next_free_vgpr', 'accum_offset')}
        print(f'{name[:70]:70s} {dict(c)} {regs}')


if __name__ == '__main__':
    main()
