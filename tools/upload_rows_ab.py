"""Row-major HOST upload (nwknn_set_base of a numpy matrix: what the operator drop-in hands over): seconds and GB/s by size, for the library
NWKNN_LIB_PATH points at (A/B: run once per build).  usage: upload_rows_ab.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from neighborhoodwatch_amd import KnnEngine
rng = np.random.default_rng(0)
eng = KnnEngine(0)
for B, d in ((10_000, 768), (1_000_000, 768), (2_000_000, 1536)):
    base = rng.random((B, d), dtype=np.float32)
    best = 1e9
    for _ in range(4):
        t = time.perf_counter(); eng.set_base(base); best = min(best, time.perf_counter() - t)
    print(f"{B} x {d} ({base.nbytes / 1e9:.2f} GB): {best * 1e3:.1f} ms  ({base.nbytes / best / 1e9:.1f} GB/s)", flush=True)
    del base
eng.close()
