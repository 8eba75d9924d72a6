"""Build liblas_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python phones-las_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.environ.get('LAS_HIP_LIB') or os.path.join(HERE, 'liblas_hip.so')
EXTRA = os.environ.get('LAS_CXXFLAGS', '').split()     # e.g. -DLAS_STAMPS for the diagnostics build
OBJDIR = os.path.join(HERE, 'build' + ('_' + '_'.join(x.strip('-') for x in EXTRA) if EXTRA else ''))


def sources():
    """HIP translation units + the plain C++ host file (csrc/host.cpp: no HIP include, so that it also builds under the
    sanitizers: tests/test_host_sanitized.py)."""
    return sorted(glob.glob(os.path.join(CSRC, '*.hip'))) + sorted(glob.glob(os.path.join(CSRC, '*.cpp')))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'las_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(OBJDIR, exist_ok=True)
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wno-unused-value'] + EXTRA + ['-c', src, '-o', obj]
        procs.append((cmd, subprocess.Popen(cmd)))
    # While the objects compile: the assembly of lstm.hip, built with the SAME flags, is scanned for compiler instructions that
    # touch the destination registers of an inline-assembly polling load before its wait (scripts/check_wide_polls.py; ADVICE r5:
    # the guard belongs to every build, diagnostics builds included, not to a test that can be skipped).  LAS_SKIP_ISA_CHECK=1 skips.
    isa = None
    if os.environ.get('LAS_SKIP_ISA_CHECK', '0') != '1':
        import importlib.util
        spec = importlib.util.spec_from_file_location('check_wide_polls', os.path.join(HERE, '..', 'scripts', 'check_wide_polls.py'))
        isa = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(isa)
        asm_path = os.path.join(OBJDIR, 'lstm.isa_check.s')
        isa.compile_to_asm(asm_path, EXTRA)
        nload, bad = isa.check(open(asm_path).read())
        os.remove(asm_path)
        if bad or nload == 0:
            for _, p in procs:
                p.kill()
            raise RuntimeError('lstm.hip: %d polling loads, %d instructions touch their destination registers in flight: %s'
                               % (nload, len(bad), bad[:3]))
        # ... and gemm.hip's las_gemm_nt_bimg kernels: hand-counted fragment loads (check_fragment_loads)
        asm_path = os.path.join(OBJDIR, 'gemm.isa_check.s')
        isa.compile_to_asm(asm_path, EXTRA, name='gemm.hip')
        nk, nfrag, gbad = isa.check_fragment_loads(open(asm_path).read())
        os.remove(asm_path)
        if gbad or nk == 0:
            for _, p in procs:
                p.kill()
            raise RuntimeError('gemm.hip: %d image-GEMM kernels, %d fragment loads, %d instructions touch a fragment in flight: %s'
                               % (nk, nfrag, len(gbad), gbad[:3]))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT]
    subprocess.check_call(cmd)
    if verbose:
        print('built', OUT, file=sys.stderr)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
