"""MaxSim ring kernel (pre-screen, d = 128): super-tile shape x walk order on the C5 shape.  Only the documents stream (the query
fragments live in registers), so the more query groups of an XCD walk the SAME document strip at the same time, the fewer bytes
cross the fabric.  Every variant's result must equal the first.  usage: ms_order_sweep.py [n_docs] [n_queries]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d, k = 128, 1000
cfgs = [(3, 0), (6, 0), (5, 0), (4, 0), (3, 1), (4, 1), (5, 1), (6, 1), (2, 0)]
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
docs = eng.generate(1, 0, B * 64, d).view(B, 64, d); q = eng.generate(2, 0, Q * 32, d).view(Q, 32, d)
eng.maxsim_set_docs(docs)
ref = None
best = {c: None for c in cfgs}
for rnd in range(2):
    for c in cfgs:
        eng.set_option("super_q_shift", c[0]); eng.set_option("super_walk_q", c[1])
        score, doc = eng.maxsim_search(q, k)
        s = eng.stats()
        if ref is None:
            ref = (score.clone(), doc.clone())
        elif not (torch.equal(ref[0], score) and torch.equal(ref[1], doc)):
            print(f"MISMATCH {c}", flush=True); sys.exit(1)
        if best[c] is None or s["tile_ms"] < best[c]["tile_ms"]:
            best[c] = s
for c in cfgs:
    s = best[c]
    print(f"{1 << c[0]:2d} query groups x {64 >> c[0]:2d} strips, {'q' if c[1] else 'b'} fastest: tile {s['tile_ms']:8.1f} ms = {s['tile_flops'] / s['tile_ms'] / 1e9:7.1f} TF  "
          f"total {s['total_ms']:8.1f} ms  x{best[cfgs[0]]['tile_ms'] / s['tile_ms']:.3f}", flush=True)
