"""MaxSim (C5 shape, pre-screened): effect of the largest chunk (option max_chunk, in token rows: 2^20 = 16384 docs) on the search time;
every variant's result must equal the first.  usage: ms_chunk_sweep.py [n_docs] [n_queries]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d, k = 128, 1000
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
docs = eng.generate(1, 0, B * 64, d).view(B, 64, d); q = eng.generate(2, 0, Q * 32, d).view(Q, 32, d)
eng.maxsim_set_docs(docs)
ref = None; best = {}
for rnd in range(3):
    for sh in (20, 21, 22, 23, 24):
        eng.set_option("max_chunk", 1 << sh)
        score, doc = eng.maxsim_search(q, k)
        s = eng.stats()
        if ref is None: ref = (score.clone(), doc.clone())
        elif not (torch.equal(ref[0], score) and torch.equal(ref[1], doc)): print("MISMATCH", sh); sys.exit(1)
        if sh not in best or s["total_ms"] < best[sh]["total_ms"]: best[sh] = s
for sh, s in best.items():
    print(f"max_chunk 2^{sh} ({(1 << sh) // 64:7d} docs): total {s['total_ms']:8.1f} ms ({Q / s['total_ms'] * 1e3:7.1f} q/s)  tile {s['tile_ms']:7.1f}  rescore {s['rescore_ms']:6.1f}  compact {s['compact_ms']:5.1f}  "
          f"launches {s['tile_launches']}  overflowed {s['overflowed_queries']} rescans {s['rescan_launches']}", flush=True)
