import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def _usable_cpus() -> int:
    """The affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 CPUs and grant the container 16)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# Before numpy / torch / the oracle load their thread pools: one thread per CPU the container may USE.  Left at their defaults (one per CPU
# shown) the checkers -- OpenMP in the oracle, BLAS behind numpy -- took 25x longer on the GPU box (68 s against 2.6 s for one test file).
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, str(_usable_cpus()))


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
