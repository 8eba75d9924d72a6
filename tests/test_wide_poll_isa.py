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
