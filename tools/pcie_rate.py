"""PCIe-inclusive rate (DESIGN note, never the bench `value`): C2 handed over as HOST numpy arrays, timed from the call of set_base to the
result arrays in host memory."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 10_000, 1_000_000, 768, 100
rng = np.random.default_rng(0)
base = rng.random((B, d), dtype=np.float32) * 2 - 1
queries = rng.random((Q, d), dtype=np.float32) * 2 - 1
cols = [np.ascontiguousarray(base[:, j]) for j in range(d)]
qcols = [np.ascontiguousarray(queries[:, j]) for j in range(d)]
for pre in (False, True):
    eng = KnnEngine(0)
    eng.set_prescreen(pre)
    for label, fn in (("row-major host arrays", lambda: (eng.set_base(base), eng.search(queries, k))),
                      ("column arrays (parquet layout)", lambda: (eng.set_base_columns(cols), eng.search_columns(qcols, k)))):
        fn()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        print(f"prescreen={int(pre)} {label}: {t1 - t0:.3f} s per job = {Q / (t1 - t0):.0f} q/s incl. H2D of {base.nbytes / 1e9:.1f} GB ({base.nbytes / 1e9 / (t1 - t0):.1f} GB/s if it were all copy)", flush=True)
    eng.close()
