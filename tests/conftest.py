import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically where no device is visible (this container).
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def pytest_sessionstart(session):
    # LAS_NANFILL=1: torch.empty() hands out NaN / max-int filled memory, so a kernel that reads what nobody wrote
    # turns results into NaN instead of into "plausible" numbers (debugging aid for allocator-dependent flakiness)
    if os.environ.get('LAS_NANFILL'):
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
