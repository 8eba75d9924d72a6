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


def check_fragment_loads(asm_text, kernel='gemm_nt_bimg_kernel'):
    """gemm.hip, las_gemm_nt_bimg: every load of the K loop is inline assembly -- `global_load_dwordx4` of a weight fragment into
    registers, `global_load_lds_dwordx4` of the activation tile -- counted by hand: a fragment is valid behind the first
    `s_waitcnt vmcnt(N)` at which at most N loads were issued after it.  Until then nothing may read or write its registers
    (a first version of the loop had guarded issues: the compiler rotated the fragment slots through v_mov copies at the
    control-flow merges, of registers whose loads had not landed).  The loop body is walked twice (the back edge).
    -> (kernels checked, fragment loads seen, violations)."""
    lines = asm_text.split('\n')
    nk, nload, bad = 0, 0, []
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w*%s\w*):' % kernel, lines[i])
        if not m:
            i += 1
            continue
        nk += 1
        end = next(j for j in range(i, len(lines)) if '.amdhsa_kernel' in lines[j] or lines[j].startswith('.Lfunc_end'))
        body = lines[i:end]
        labels = {re.match(r'^(\.LBB\w+):', l).group(1): k for k, l in enumerate(body) if re.match(r'^\.LBB\w+:', l)}
        issued, pending = 0, []          # pending: (sequence number, registers)

        def walk(a, b):
            nonlocal issued, pending, nload
            inasm = False
            for k in range(a, b):
                t = body[k].strip()
                if t.startswith(';;#ASMSTART'):
                    inasm = True
                    continue
                if t.startswith(';;#ASMEND'):
                    inasm = False
                    continue
                if not t or t[0] in ';.' or t.endswith(':'):
                    continue
                t = t.split(';')[0].strip()
                mm = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off$', t)
                if inasm and mm:
                    issued += 1
                    nload += 1
                    pending.append((issued, set(range(int(mm.group(1)), int(mm.group(2)) + 1))))
                    continue
                if inasm and t.startswith('global_load_lds_dwordx4'):
                    issued += 1
                    continue
                mm = re.match(r's_waitcnt vmcnt\((\d+)\)', t)
                if mm:
                    pending = [(q, r) for q, r in pending if q > issued - int(mm.group(1))]
                    continue
                regs = set()
                for x, y in re.findall(r'\bv\[(\d+):(\d+)\]', t):
                    regs.update(range(int(x), int(y) + 1))
                regs.update(int(x) for x in re.findall(r'\bv(\d+)\b', t))
                for q, r in pending:
                    if regs & r:
                        bad.append((i + k + 1, t, sorted(regs & r)))
        back = None
        for k, l in enumerate(body):
            mm = re.match(r'\s*s_cbranch_\w+ (\.LBB\w+)\s*$', l.split(';')[0])
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k and any('vmcnt(12)' in x for x in body[labels[mm.group(1)]:k]):
                back = (labels[mm.group(1)], k)
        if back is None:
            bad.append((i + 1, 'no K loop found in ' + m.group(1), []))
        else:
            walk(0, back[1])
            walk(back[0], back[1])          # once more around the back edge
            walk(back[1], len(body))
        i = end
    return nk, nload, bad


def compile_to_asm(out_path, extra=None, name='lstm.hip'):
    """extra: the additional compiler flags of the build being checked (default: LAS_CXXFLAGS, as build.py reads it)."""
    src = os.path.join(ROOT, 'phones-las_amd', 'csrc', name)
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
        gpath = os.path.join(d, 'gemm.s')
        compile_to_asm(gpath, name='gemm.hip')
        nk, nfrag, gbad = check_fragment_loads(open(gpath).read())
    print('inline-assembly wide loads: %d; compiler instructions that touch a destination register in flight: %d' % (nload, len(bad)))
    print('las_gemm_nt_bimg: %d kernels, %d fragment loads walked; instructions that touch a fragment in flight: %d' % (nk, nfrag, len(gbad)))
    for ln, ins, regs in (bad + gbad)[:10]:
        print('  line %d: %s   (v%s)' % (ln, ins, regs))
    return 1 if bad or nload == 0 or gbad or nk == 0 else 0


if __name__ == '__main__':
    sys.exit(main())
