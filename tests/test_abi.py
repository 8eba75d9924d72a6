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
