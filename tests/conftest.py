import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Test aid of libcrgpu.so's buffer pool (csrc/runtime.hip, DevBuf::ensure): every block a buffer receives is filled with
# this byte first.  Without it a pooled block keeps what its previous owner wrote, and a kernel that forgets part of its
# output (e.g. the "present" bytes of new groups -- it happened) can pass by luck.  0 exposes missing non-zero writes;
# run the suite with CRGPU_POOL_POISON=165 as well to expose outputs that rely on fresh memory being zero.
os.environ.setdefault("CRGPU_POOL_POISON", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


def pytest_collection_modifyitems(config, items):
    """Some GPU tests build their inputs with torch.  torch bundles its own HIP runtime, which only finds the
    device if it initialises BEFORE libcrgpu.so's first HIP call in this process (bench.py has the same order),
    so when GPU tests were selected, torch touches the device first."""
    if not any(it.get_closest_marker("gpu") for it in items):
        return
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
