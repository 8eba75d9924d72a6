"""Build-time guard for the inline-assembly polls of the forward recurrence (scripts/check_wide_polls.py): in the gfx950 assembly of
lstm.hip no compiler-emitted instruction may touch the destination registers of a 16-byte polling load before the `s_waitcnt` that
follows the round.  Runs hipcc (cross-compilation: no GPU), about a minute."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location('check_wide_polls', os.path.join(ROOT, 'scripts', 'check_wide_polls.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_checker_sees_a_planted_violation():
    chk = _checker()
    good = '\n'.join([';;#ASMSTART', 'global_load_dwordx4 v[10:13], v[2:3], off sc1', ';;#ASMEND', 'v_mov_b32_e32 v20, 1',
                      ';;#ASMSTART', 's_waitcnt vmcnt(0)', ';;#ASMEND', 'v_add_u32_e32 v21, v11, v20'])
    assert chk.check(good) == (1, [])
    bad = good.replace('v_mov_b32_e32 v20, 1', 'v_mov_b32_e32 v20, v12')
    n, found = chk.check(bad)
    assert n == 1 and len(found) == 1 and found[0][2] == [12]


@pytest.mark.skipif(shutil.which(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')) is None, reason='hipcc not installed')
def test_no_instruction_touches_a_polling_load_in_flight(tmp_path):
    chk = _checker()
    path = str(tmp_path / 'lstm.s')
    chk.compile_to_asm(path)
    nload, bad = chk.check(open(path).read())
    assert nload >= 100, nload                 # every exchanging instantiation of the forward kernel has them
    assert not bad, bad[:5]


def _bimg_asm(rotate):
    """A toy las_gemm_nt_bimg kernel: three fragment slots (v[10:13], v[14:17], v[18:21]); a prologue that issues the fragment and the
    LDS-DMA load of stages 0..2; a K loop written out three stages at a time, each of which waits vmcnt(4) (= everything but the two
    stages behind it in the queue), multiplies with its slot and re-issues the slot for the stage three on.  rotate: a copy of the
    NEXT slot (still in flight) behind the first stage -- what the first version of the kernel did at its control-flow merges."""
    def issue(r):
        return [';;#ASMSTART', 'global_load_dwordx4 v[%d:%d], v[2:3], off' % (r, r + 3), ';;#ASMEND', ';;#ASMSTART', 'global_load_lds_dwordx4 v[4:5], off', ';;#ASMEND']
    body = ['_Z19gemm_nt_bimg_kernelILi4ELi8EEv8GemmArgs:'] + issue(10) + issue(14) + issue(18) + ['.LBB0_1:                    ; =>This Inner Loop Header']
    for k, r in enumerate((10, 14, 18)):
        body += [';;#ASMSTART', 's_waitcnt vmcnt(4)', ';;#ASMEND', 'v_mfma_f32_16x16x32_bf16 v[30:33], v[40:43], v[%d:%d], v[30:33]' % (r, r + 3)] + issue(r)
        if k == 0:
            body += ['v_mov_b64_e32 v[50:51], v[14:15]' if rotate else 's_nop 1', '; vmcnt(12) is what the real kernel waits for']
    return '\n'.join(body + ['s_cbranch_scc0 .LBB0_1', 's_endpgm', '.amdhsa_kernel x'])


def test_the_fragment_load_checker_counts_the_queue():
    """check_fragment_loads (las_gemm_nt_bimg): a fragment is valid behind the first vmcnt(N) with at most N loads issued after it --
    reading v[10:13] behind `vmcnt(4)` with two stages (4 loads) behind it is fine, a copy of v[14:15] (the NEXT stage's fragment,
    still in flight) is what the first version of the kernel did at its control-flow merges and must be reported."""
    chk = _checker()
    nk, nfrag, bad = chk.check_fragment_loads(_bimg_asm(False))
    assert nk == 1 and nfrag == 3 + 2 * 3 and bad == []
    nk, nfrag, bad = chk.check_fragment_loads(_bimg_asm(True))
    assert nk == 1 and bad and all(b[2] == [14, 15] for b in bad)


@pytest.mark.skipif(shutil.which(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')) is None, reason='hipcc not installed')
def test_no_instruction_touches_a_weight_fragment_in_flight(tmp_path):
    chk = _checker()
    path = str(tmp_path / 'gemm.s')
    chk.compile_to_asm(path, name='gemm.hip')
    nk, nfrag, bad = chk.check_fragment_loads(open(path).read())
    assert nk == 3 and nfrag >= 3 * 28, (nk, nfrag)        # three tile heights; 12 loads in the prologue + 16 in the loop, each
    assert not bad, bad[:5]
