import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def engine():
    """One HIP engine for the whole GPU session (tests call through the C ABI via this wrapper)."""
    import torch
    torch.cuda.init()                     # torch's HIP context first, then ours, in one shared runtime (see _lib.py)
    from neighborhoodwatch_amd import KnnEngine
    eng = KnnEngine(0)
    yield eng
    eng.close()
