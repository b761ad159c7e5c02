"""MaxSim pre-screen tile kernel, d = 128: ring form (ms_ring=1; ms_mfma 0 = 32x32x16, 1 = 16x16x32) against the strip kernel (ms_ring=0) over strip lengths, interleaved in
one process on the C5 shape (2048 queries x 32 tokens, 1M docs x 64 tokens, k = 1000); every variant's result must equal the first.
usage: ms_ring_ab.py [n_docs] [n_queries]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d, k = 128, 1000
variants = [(1, 0, 0), (1, 0, 1), (0, 8, 0), (1, 8, 0), (1, 8, 1), (1, 32, 1)]      # (ms_ring, ms_strip (0 = auto: 64), ms_mfma (1 = 16x16x32))
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
docs = eng.generate(1, 0, B * 64, d).view(B, 64, d); q = eng.generate(2, 0, Q * 32, d).view(Q, 32, d)
eng.maxsim_set_docs(docs)
ref = None
best = {v: None for v in variants}
for rnd in range(3):
    for v in variants:
        eng.set_option("ms_ring", v[0]); eng.set_option("ms_strip", v[1]); eng.set_option("ms_mfma", v[2])
        score, doc = eng.maxsim_search(q, k)
        s = eng.stats()
        if ref is None:
            ref = (score.clone(), doc.clone())
        elif not (torch.equal(ref[0], score) and torch.equal(ref[1], doc)):
            print(f"MISMATCH ring={v[0]} strip={v[1]} mfma={v[2]}", flush=True)
            sys.exit(1)
        if best[v] is None or s["tile_ms"] < best[v]["tile_ms"]:
            best[v] = s
for v in variants:
    s = best[v]
    print(f"ring={v[0]} strip={v[1]:2d} mfma={v[2]}: tile {s['tile_ms']:8.1f} ms = {s['tile_flops'] / s['tile_ms'] / 1e9:7.1f} TF   rescore {s['rescore_ms']:6.1f} ms   total {s['total_ms']:8.1f} ms "
          f"({Q / s['total_ms'] * 1e3:7.1f} q/s)   x{best[variants[0]]['tile_ms'] / s['tile_ms']:.3f}", flush=True)
