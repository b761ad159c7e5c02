"""Is the fp16 tile kernel power-limited?  Same launches on random and on all-zero operands (zero data toggles nothing: the chip
keeps its clock); the tile kernel's TFLOP/s on both."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 32768, 1 << 20, 1024, 100
eng = KnnEngine(0, profile=True)
eng.set_compute_dtype("fp16")
for name in ("random", "zeros", "random", "zeros"):
    base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
    if name == "zeros":
        base.zero_(); q.zero_()
    eng.set_base(base)
    best = 0.0
    for _ in range(3):
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best = max(best, s["tile_flops"] / s["tile_ms"] / 1e9)
    print(f"{name:7s}: tile kernel {best:7.1f} TF", flush=True)
