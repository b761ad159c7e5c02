"""Per-tile cost model of the fp16 tile kernel: tile_ms at several d (same Q, B) -> fixed cost per tile + cost per 64-half slab.
usage: python tools/f16_fit.py [prescreen|f16] [f16_mfma]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from neighborhoodwatch_amd import KnnEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
shape = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q, B, k = 32768, 1 << 20, 100
eng = KnnEngine(0, profile=True)
eng.set_option("f16_mfma", shape)
if mode == "prescreen":
    eng.set_prescreen(True)
else:
    eng.set_compute_dtype("fp16")
rows = []
for d in (256, 512, 1024, 2048, 4096):
    base = eng.generate(1, 0, B, d)
    q = eng.generate(2, 0, Q, d)
    eng.set_base(base)
    best = None
    for _ in range(3):
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best = s if best is None or s["tile_ms"] < best["tile_ms"] else best
    tiles = (Q / 256) * (B / 256)
    us_per_tile = best["tile_ms"] * 1e3 / tiles * 256        # x256 CUs: wall time a CU spends per tile
    rows.append((d, best["tile_ms"], best["tile_flops"] / best["tile_ms"] / 1e9, us_per_tile))
    print(f"d={d:5d} tile_ms={best['tile_ms']:8.2f}  {rows[-1][2]:7.1f} TF  {us_per_tile:6.2f} us/tile/CU", flush=True)
    del base, q
x = np.array([r[0] / 64 for r in rows]); y = np.array([r[3] for r in rows])
b, a = np.polyfit(x, y, 1)
print(f"fit: {a:.2f} us fixed per tile + {b:.3f} us per 64-half slab  -> asymptotic {2*256*256*64/ (b*1e-6) * 256 / 1e12:.0f} TF; fixed = {a/b:.1f} slabs")
