"""Super-tile shape x walk order (super_walk_q 1 = query tile index fastest) for the direct fp32 kernel and the pre-screen kernel.
usage: order_sweep.py [direct|pre] [Q] [B] [d]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
mode = sys.argv[1] if len(sys.argv) > 1 else "pre"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
k = 100
eng = KnnEngine(0, profile=True)
eng.set_prescreen(mode == "pre")
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
cfgs = [(3, 0), (3, 4), (2, 0), (2, 4), (1, 0), (1, 4), (4, 0), (4, 4), (0, 4), (5, 0)]
best = {c: 0.0 for c in cfgs}
for rnd in range(2):
    for c in cfgs:
        eng.set_option("super_q_shift", c[0]); eng.set_option("super_walk_q", 1 if c[1] else 0)
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best[c] = max(best[c], s["tile_flops"] / s["tile_ms"] / 1e9)
print(f"{mode} Q={Q} B={B} d={d}")
for c in cfgs:
    print(f"  {1 << c[0]:2d} q-tiles x {64 >> c[0]:2d} b-tiles, {'q' if c[1] else 'b'} fastest: {best[c]:7.1f} TF ({best[c] / best[cfgs[0]]:.3f}x)", flush=True)
