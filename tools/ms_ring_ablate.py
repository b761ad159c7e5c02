"""Where does the MaxSim ring kernel (pre-screen, d = 128) spend its time?  Timing ablations behind ms_ring = 3 (no document stream inside
the loop: the MFMAs run on whatever the ring holds), 5 (epilogue only for the last tile of a strip), 7 (both); results are wrong by
construction, only tile_ms is read.  C5 shape.  usage: ms_ring_ablate.py [n_docs] [n_queries]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d, k = 128, 1000
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
docs = eng.generate(1, 0, B * 64, d).view(B, 64, d); q = eng.generate(2, 0, Q * 32, d).view(Q, 32, d)
eng.maxsim_set_docs(docs)
best = {}
for rnd in range(2):
    for v in (1, 3, 5, 7):
        eng.set_option("ms_ring", v)
        eng.maxsim_search(q, k)
        s = eng.stats()
        best[v] = min(best.get(v, 1e30), s["tile_ms"])
        print(v, {key: s[key] for key in ("tile_ms", "tile_launches", "rescan_launches", "overflowed_queries", "rescore_ms")}, flush=True)
fl = 2.0 * Q * 32 * B * 64 * d
for v, name in ((1, "nothing removed"), (3, "no document stream"), (5, "no epilogue"), (7, "neither")):
    print(f"ms_ring={v} ({name:18s}): tile {best[v]:8.1f} ms = {fl / best[v] / 1e9:7.1f} TF  x{best[1] / best[v]:.3f}", flush=True)
