"""C2 (10k x 1M, d = 768, k = 100, fp32 direct): tile shape x workgroups per CU x chunk growth, interleaved; results must be identical.
usage: c2_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 10_000, 1_000_000, 768, 100
eng = KnnEngine(0, profile=True)
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
cfgs = [(0, 3, 0), (256, 3, 0), (256, 2, 0), (128, 3, 0), (128, 2, 0), (0, 3, 2), (0, 3, 5), (0, 3, 7)]
ref = None; best = {c: None for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        eng.set_option("tile_rows", c[0]); eng.set_option("tile_occ", c[1]); eng.set_option("chunk_growth", c[2])
        dist, idx = eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        if ref is None: ref = (dist.clone(), idx.clone())
        elif not (torch.equal(ref[0], dist) and torch.equal(ref[1], idx)): print("MISMATCH", c); sys.exit(1)
        if best[c] is None or s["total_ms"] < best[c]["total_ms"]: best[c] = s
for c in cfgs:
    s = best[c]
    print(f"tile_rows {c[0]:3d} occ {c[1]} growth {c[2]}: total {s['total_ms']:7.2f} ms ({Q / s['total_ms'] * 1e3:8.0f} q/s) tile {s['tile_ms']:7.2f} ms = {s['tile_flops'] / s['tile_ms'] / 1e9:6.1f} TF, "
          f"{s['tile_launches']} launches, compact {s['compact_ms']:.2f} prep {s['prep_ms']:.2f}", flush=True)
