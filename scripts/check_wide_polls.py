"""The forward recurrence's 16-byte polls are inline assembly: a `global_load_dwordx4 ... sc1` per pair of granules and one
`s_waitcnt vmcnt(0)` behind the round (phones-las_amd/csrc/lstm.hip, wide_round).  The compiler does not know that the destination
registers of such a load are written LATER, by the memory system: nothing it emits between the load and the wait may read or write
them.  This script compiles lstm.hip to gfx950 assembly and checks exactly that for every instantiation (no GPU needed):
    python scripts/check_wide_polls.py        -> prints the count, exit code 1 on a violation"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    """-> (number of inline-assembly wide loads, list of (line number, instruction, registers) violations)."""
    pending, bad, nload, inasm = set(), [], 0, False
    for i, line in enumerate(asm_text.split('\n')):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        m = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\]|v\d+), (off|s\[\d+:\d+\]) sc1', t)
        if inasm and m:
            nload += 1
            pending.update(range(int(m.group(1)), int(m.group(2)) + 1))
            continue
        if inasm and t.startswith('s_waitcnt vmcnt(0)'):
            pending = set()
            continue
        if not pending or not t or t[0] in ';.' or t.endswith(':'):
            continue
        regs = set()
        for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', t):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r'\bv(\d+)\b', t))
        if regs & pending:
            bad.append((i + 1, t, sorted(regs & pending)))
    return nload, bad


def compile_to_asm(out_path, extra=None):
    """extra: the additional compiler flags of the build being checked (default: LAS_CXXFLAGS, as build.py reads it)."""
    src = os.path.join(ROOT, 'phones-las_amd', 'csrc', 'lstm.hip')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if extra is None:
        extra = os.environ.get('LAS_CXXFLAGS', '').split()
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-Wno-unused-value'] + list(extra) +
                   ['-I', os.path.dirname(src), src, '-o', out_path], check=True, stderr=subprocess.DEVNULL)


def main():
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'lstm.s')
        compile_to_asm(path)
        nload, bad = check(open(path).read())
    print('inline-assembly wide loads: %d; compiler instructions that touch a destination register in flight: %d' % (nload, len(bad)))
    for ln, ins, regs in bad[:10]:
        print('  line %d: %s   (v%s)' % (ln, ins, regs))
    return 1 if bad or nload == 0 else 0


if __name__ == '__main__':
    sys.exit(main())
