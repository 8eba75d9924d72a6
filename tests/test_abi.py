"""The C-ABI library loads on a machine without a GPU and exports every symbol include/las_hip.h declares;
the ctypes binding covers each of them.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'las_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(las_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    names = _declared()
    assert 'las_gemm_nt' in names and 'las_lstm_recurrent_fwd' in names and 'las_adam_update' in names
    assert len(names) >= 18


def test_library_exports_every_declared_symbol():
    from phones_las_amd import hip
    if not os.path.exists(hip.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(hip.lib_path())
    for name in _declared():
        assert hasattr(lib, name), name


def test_binding_covers_header_and_nothing_pending():
    from phones_las_amd import hip
    assert hip._PENDING == set()
    assert sorted(hip.EXPORTS) == _declared()
    l = hip.lib()
    assert l.las_version() >= 100
    assert l.las_lstm_workspace_bytes(64, 256, 2) > 0
    assert l.las_lstm_workspace_bytes(64, 100, 2) == 0        # unsupported num_units -> 0, no crash


def test_ops_refuse_cpu_tensors():
    import torch
    from phones_las_amd import hip
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(hip.LasError):
        hip.gemm_nt(a, a, torch.zeros(8, 8), 8, 8, 8)


def test_slice_height_policy_and_workspace_sizes(monkeypatch):
    """Host-side queries of the recurrent kernels (no GPU work): utterances per slice as a function of the batch
    (every chain workgroup and companion keeps a CU of its own out of 256), the override, and a workspace that covers
    every layout."""
    from phones_las_amd import hip
    l = hip.lib()
    hip.set_knob('LAS_LSTM_ROWS', 0)
    assert l.las_lstm_slice_rows(64, 256, 2) == 4            # 32 chains x 4 workgroups + 32 companions = 160
    assert l.las_lstm_slice_rows(96, 256, 2) == 4            # 48 x 4 + 48 = 240
    assert l.las_lstm_slice_rows(128, 256, 2) == 8           # 4-row slices would need 320 workgroups
    assert l.las_lstm_slice_rows(512, 256, 2) == 16
    assert l.las_lstm_slice_rows(64, 512, 2) == 8            # 512 units as 8 members (round 3): 16 chains x 8 + 32 companions = 160
    assert l.las_lstm_slice_rows(16, 512, 2) == 4            # 8 chains x 8 + 16 = 80
    assert l.las_lstm_slice_rows(64, 1024, 2) == 8           # 32-unit members (K / row split): half or full tiles only
    assert l.las_lstm_slice_rows(8, 128, 2) == 4
    assert l.las_lstm_slice_rows(0, 256, 2) == 0 and l.las_lstm_slice_rows(8, 100, 2) == 0
    hip.set_knob('LAS_LSTM_ROWS', 16)
    assert l.las_lstm_slice_rows(64, 256, 2) == 16
    hip.set_knob('LAS_LSTM_ROWS', 0)
    # the workspace covers the forward and the backward exchange of every slice height
    w = l.las_lstm_workspace_bytes(64, 256, 2)
    assert w >= 64 + 2 * 32 * 4 * 4 * 4 * 256 * 8            # backward, 4-row slices: 32 groups x 4 x 4 pairs x NUB*256 granules x 2 slots
    # one workgroup per chain: nothing to exchange, but the chain's progress granules for its prefetch companion live there
    assert 64 < l.las_lstm_workspace_bytes(64, 128, 2) <= 64 + 4 * 1024 * 1024


def test_ctypes_structures_have_the_layout_of_the_header(tmp_path):
    """Every struct the C-ABI takes by pointer: size and the offset of each field of the ctypes mirror against the header as gcc
    lays it out (a field added on one side only would silently shift everything behind it)."""
    import subprocess
    from phones_las_amd import hip
    pairs = {'las_lstm_fwd': hip.LstmFwd, 'las_dec_step': hip.DecStep, 'las_dec_persist': hip.DecPersist,
             'las_dec_step_bwd': hip.DecStepBwd, 'las_dec_persist_bwd': hip.DecPersistBwd, 'las_dec_seq_bwd': hip.DecSeqBwd,
             'las_image_job': hip.ImageJob, 'las_fill_job': hip.FillJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "las_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s . %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        c, f, v = line.split()
        got[(c, f)] = int(v)
    for cname, cls in pairs.items():
        assert got[(cname, '.')] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
