#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel of the library for instructions that have no business in it.

    python scripts/check_isa.py [--compile] [file.s ...]

`flat_*` memory instructions: the address space of an access was lost in a cast (round 6: the recurrent backward kernels read
their LDS timeout flag through a generic `volatile int*`, which became `flat_load_dword ... sc0 sc1` + `s_waitcnt vmcnt(0)`
behind the per-step barrier -- the chain waited there for the acknowledgement of its HBM stores).  `scratch_*`: register spills.
--compile builds the assembly of phones-las_amd/csrc/*.hip into a temporary directory first (hipcc cross-compiles without a GPU;
LAS_CXXFLAGS is honoured, so a diagnostics build is checked with the flags it is built with).
Prints one line per kernel that has any; exit status 1 when a kernel outside ALLOWED has flat instructions.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'phones-las_amd', 'csrc')
# kernels that may use flat instructions (generic pointers by design); regular expressions on the mangled name
ALLOWED = []


def compile_all(outdir):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('LAS_CXXFLAGS', '').split()
    procs = []
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        out = os.path.join(outdir, os.path.basename(src)[:-4] + '.s')
        cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-Wno-unused-value', '-S', '--cuda-device-only'] + extra + [src, '-o', out]
        procs.append((out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    outs = []
    for out, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed for ' + out)
        outs.append(out)
    return outs


def scan(path):
    text = open(path).read()
    rows = []
    for m in re.finditer(r'^(\w+):\s*; @\1\n', text, re.M):
        name = m.group(1)
        end = text.find('.Lfunc_end', m.end())
        body = text[m.end():end if end > 0 else len(text)]
        if '.amdhsa_kernel ' + name not in text:
            continue                                   # a device function, not a kernel
        flat = len(re.findall(r'^\s*flat_(load|store|atomic)', body, re.M))
        scratch = len(re.findall(r'^\s*scratch_(load|store)', body, re.M))
        if flat or scratch:
            rows.append((os.path.basename(path), name, flat, scratch))
    return rows


def main(argv):
    files = [a for a in argv if not a.startswith('--')]
    tmp = None
    if '--compile' in argv or not files:
        tmp = tempfile.mkdtemp(prefix='las_isa_')
        files = compile_all(tmp)
    bad = 0
    for f in files:
        for fname, name, flat, scratch in scan(f):
            ok = any(re.search(p, name) for p in ALLOWED)
            print('%-14s flat %3d  scratch %4d  %s%s' % (fname, flat, scratch, name[:150], '' if (ok or not flat) else '   <-- FLAT'))
            if flat and not ok:
                bad += 1
    print('%d kernel(s) with unexpected flat instructions' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
