"""Per-tile cost model of the fp32 tile kernel: tile_ms at several d (same Q, B) -> fixed cost per 256x128 tile + cost per unit of d."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from neighborhoodwatch_amd import KnnEngine
Q, B, k = 32768, 1 << 19, 100
eng = KnnEngine(0, profile=True)
rows = []
for d in (128, 256, 512, 1024, 2048):
    base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
    eng.set_base(base)
    best = None
    for _ in range(2):
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best = s if best is None or s["tile_ms"] < best["tile_ms"] else best
    tiles = (Q / 256) * (B / 128)
    us = best["tile_ms"] * 1e3 / tiles * 512                 # 2 workgroups per CU x 256 CUs: residency per tile
    rows.append((d, us))
    print(f"d={d:5d} tile_ms={best['tile_ms']:8.2f}  {best['tile_flops'] / best['tile_ms'] / 1e9:7.1f} TF  {us:7.2f} us/tile (per workgroup slot)", flush=True)
    del base, q
x = np.array([r[0] for r in rows], dtype=float); y = np.array([r[1] for r in rows])
b, a = np.polyfit(x, y, 1)
print(f"fit: {a:.2f} us fixed per tile + {b*1024:.2f} us per 1024 dims -> asymptotic {2*256*128*1024/(b*1024*1e-6)*512/1e12:.1f} TF; fixed = {a/(b*1024)*100:.1f} % of a d=1024 tile")
