"""Column upload (nwknn_set_base_columns) time vs host threads, C2-sized base held in numpy column arrays.  usage: upload_times.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from neighborhoodwatch_amd import KnnEngine
B, d = 1_000_000, 768
rng = np.random.default_rng(0)
cols = [rng.random(B, dtype=np.float32) for _ in range(d)]
eng = KnnEngine(0)
for nt in (8, 1, 2, 4, 8, 16):
    eng.set_option("upload_threads", nt)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); eng.set_base_columns(cols); best = min(best, time.perf_counter() - t)
    print(f"threads {nt:2d}: {best:.3f} s  ({B * d * 4 / best / 1e9:.1f} GB/s)", flush=True)
