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
    config.addinivalue_line("markers", "drives_fallback: the test provokes the pre-screen's fallback on purpose (the invariant hook below stands aside)")


@pytest.fixture(autouse=True)
def _prescreen_invariant_holds_after_every_search(request, monkeypatch):
    """Every kNN search of every GPU test: the invariant behind the fp16 pre-screen (|estimate - exact| <= the bound pairs were pruned
    under, counted on the device over all rescored pairs: nwknn_stats.prescreen_*) must have held -- no violation, no fallback."""
    if "gpu" not in request.keywords or request.node.get_closest_marker("drives_fallback"):
        yield
        return
    from neighborhoodwatch_amd import engine as E

    def checked(orig):
        def call(self, *a, **kw):
            out = orig(self, *a, **kw)
            st = self.stats()
            assert st["prescreen_violations"] == 0 and st["prescreen_fallbacks"] == 0 and st["prescreen_worst"] < 1.0, st
            return out
        return call
    monkeypatch.setattr(E.KnnEngine, "search", checked(E.KnnEngine.search))
    monkeypatch.setattr(E.KnnEngine, "search_columns", checked(E.KnnEngine.search_columns))
    monkeypatch.setattr(E.KnnEngine, "maxsim_search", checked(E.KnnEngine.maxsim_search))
    yield


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
