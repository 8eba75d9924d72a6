#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy of every kernel of liblas_hip as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).  Prints one line per kernel instantiation, the ones
with scratch (spills) first.  `python scripts/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt`."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return [o.replace('(anonymous namespace)::', '') for o in out[:len(names)]]
    except OSError:
        return names


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'phones-las_amd', 'csrc', '*.hip')))
    rows = []
    for f in files:
        with tempfile.NamedTemporaryFile(suffix='.o') as o:
            r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wno-unused-value',
                                '-Rpass-analysis=kernel-resource-usage', '-c', f, '-o', o.name], capture_output=True, text=True)
        for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            name = b.split('\n')[0].split(' ')[0]
            g = lambda k: int((re.search(k + r": (\d+)", b) or [0, 0])[1])
            rows.append((os.path.basename(f), name, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
                         g(r'LDS Size \[bytes/block\]'), g('SGPRs')))
    names = demangle([r[1] for r in rows])
    rows = [r[:1] + (n,) + r[2:] for r, n in zip(rows, names)]
    rows.sort(key=lambda r: (-r[4], r[0], r[1]))
    print('%-13s %5s %5s %8s %4s %7s %5s  kernel' % ('file', 'VGPR', 'AGPR', 'scratch', 'occ', 'LDS', 'SGPR'))
    for f, n, v, a, s, occ, lds, sg in rows:
        print('%-13s %5d %5d %8d %4d %7d %5d  %s' % (f, v, a, s, occ, lds, sg, n.split('(')[0][:150]))
    print('# %d kernel instantiations, %d with scratch' % (len(rows), sum(1 for r in rows if r[4] > 0)))


if __name__ == '__main__':
    main()
