"""Layout + norm pass (nwk_prep_tiled_kernel) on the C3 base (10M x 1024 float32) and the C4 slice (12.5M x 1536 float16): wall time of
KnnEngine.set_base (device input, synchronous) -> bytes read + written per second.  usage: prep_rate.py [rounds]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, rows, d, dtype, es in (("C3 base f32", 10_000_000, 1024, "f32", 4), ("C4 slice f16", 12_500_000, 1536, "f16", 2)):
    eng = KnnEngine(0, profile=True)
    base = eng.generate(1, 0, rows, d, dtype)
    best = 1e30
    for _ in range(rounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.set_base(base)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    gb = 2.0 * rows * d * es / 1e9
    print(f"{name}: {rows} x {d}: set_base {best * 1e3:7.2f} ms -> {gb / best / 1e3:5.2f} TB/s (read + write {gb:.1f} GB)", flush=True)
    eng.close(); del base
