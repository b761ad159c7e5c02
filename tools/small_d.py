"""fp16 tile kernel builds at small d (pre-screen path, one_minus_dot): 16x16x32 8-wave kernel (strips) vs two 4-wave workgroups per CU."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
Q, B, k = 32768, 1 << 21, 100
for d in (64, 128, 192, 256, 384):
    line = f"d={d:4d}:"
    for shape in (1, 2):
        eng = KnnEngine(0, profile=True)
        eng.set_prescreen(True)
        eng.set_option("f16_mfma", shape)
        base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
        eng.set_base(base)
        best = 1e30; tot = 1e30
        for _ in range(3):
            eng.search(q, k, "one_minus_dot")
            s = eng.stats()
            best = min(best, s["tile_ms"]); tot = min(tot, s["total_ms"])
        line += f"  f16_mfma={shape}: tile {best:7.2f} ms total {tot:7.2f} ms"
        eng.close()
    print(line, flush=True)
