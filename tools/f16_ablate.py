"""Where does the fp16 tile kernel's K loop lose its time?  The 32x32x16 build (f16_mfma=0) has timing ablations behind the dev
option `diag` (wrong results by construction): 1 no fragment reads from LDS, 2 no LDS-DMA, 4 no barriers, 16 no epilogue.
Interleaved rounds in one process, random data, C3-shaped d = 1024."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 32768, 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 100
DIAGS = [0, 1, 2, 4, 16, 7, 23]
eng = KnnEngine(0, profile=True)
eng.set_compute_dtype("fp16")
eng.set_option("f16_mfma", 0)
eng.set_option("f16_exact", 0)
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
best = {x: 0.0 for x in DIAGS}
for rnd in range(3):
    for x in DIAGS:
        eng.set_option("diag", x)
        eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best[x] = max(best[x], s["tile_flops"] / s["tile_ms"] / 1e9)
for x in DIAGS:
    print(f"d={d} diag {x:2d}: tile kernel {best[x]:7.1f} TF  ({best[x] / best[0]:.3f}x)", flush=True)
