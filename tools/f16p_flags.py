"""streaming fp16 tile kernel (f16_mfma=3): dev_flags timing experiments (2 = L2-resident operand stream; results wrong), interleaved.
usage: f16p_flags.py "<flags...>" [d] [B]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
flags = [int(x) for x in sys.argv[1].split()]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
Q, k = 32768, 100
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
eng.set_option("f16_mfma", 3)
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
best = {x: 0.0 for x in flags}
for rnd in range(3):
    for x in flags:
        eng.set_option("dev_flags", x)
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best[x] = max(best[x], s["tile_flops"] / s["tile_ms"] / 1e9)
print(f"d={d} B={B}: " + "  ".join(f"flags {x}: {best[x]:7.1f} TF ({best[x] / best[flags[0]]:.3f}x)" for x in flags), flush=True)
