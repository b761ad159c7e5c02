"""streaming fp16 tile kernel (f16_mfma=3): strip length sweep (option f16_strip), interleaved.  usage: f16p_strip.py [d] [B]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
Q, k = 32768, 100
strips = [4, 2, 8, 16, 32]
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
eng.set_option("f16_mfma", 3)
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
best = {x: 0.0 for x in strips}
for rnd in range(3):
    for x in strips:
        eng.set_option("f16_strip", x)
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best[x] = max(best[x], s["tile_flops"] / s["tile_ms"] / 1e9)
print(f"d={d} B={B}: " + "  ".join(f"strip {x}: {best[x]:7.1f} TF ({best[x] / best[strips[0]]:.3f}x)" for x in strips), flush=True)
