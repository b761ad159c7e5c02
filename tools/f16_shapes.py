"""fp16 tile kernel builds against each other (option f16_mfma), pre-screen of an fp32 search, interleaved rounds in one process; the
results of every build must be the same bits.  usage: f16_shapes.py "<shapes...>" [d] [B] [Q]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
shapes = [int(x) for x in sys.argv[1].split()]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
k = 100
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
best = {x: 0.0 for x in shapes}; tot = {x: 1e30 for x in shapes}
ref = None
for rnd in range(3):
    for x in shapes:
        eng.set_option("f16_mfma", x)
        dist, idx = eng.search(q, k, "sqeuclidean")
        s = eng.stats()
        best[x] = max(best[x], s["tile_flops"] / s["tile_ms"] / 1e9); tot[x] = min(tot[x], s["total_ms"])
        if ref is None:
            ref = (dist.clone(), idx.clone())
        elif not (torch.equal(ref[0], dist) and torch.equal(ref[1], idx)):
            print(f"!! f16_mfma={x}: result differs from f16_mfma={shapes[0]}: {(ref[1] != idx).sum().item()} ids", flush=True)
print(f"d={d} B={B} Q={Q}: " + "  ".join(f"shape {x}: {best[x]:7.1f} TF total {tot[x]:8.2f} ms ({best[x] / best[shapes[0]]:.3f}x)" for x in shapes), flush=True)
