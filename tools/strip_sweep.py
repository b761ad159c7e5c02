"""fp16 tile kernel (pre-screen path): effect of f16_strip at small d, where a tile is a few slabs and the fixed per-tile cost dominates."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
Q, B, k = 32768, 1 << 21, 100
for d in (64, 128, 256, 512):
    line = f"d={d:4d}:"
    for strip in (1, 2, 4, 8, 16):
        eng = KnnEngine(0, profile=True)
        eng.set_prescreen(True)
        eng.set_option("f16_strip", strip)
        base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
        eng.set_base(base)
        best = 1e30; tot = 1e30
        for _ in range(3):
            eng.search(q, k, "one_minus_dot")
            s = eng.stats()
            best = min(best, s["tile_ms"]); tot = min(tot, s["total_ms"])
        line += f"  strip {strip:2d}: tile {best:7.2f} ms total {tot:7.2f} ms"
        eng.close()
    print(line, flush=True)
