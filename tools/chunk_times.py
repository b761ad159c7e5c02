"""Per-launch time of the tile kernel over the chunk walk of one search (run under rocprofv3 --kernel-trace), for one 8-GPU
shard of C3: 100k queries x 1.25M base rows, d = 1024, pre-screened.  usage: chunk_times.py [B] [prescreen 0|1]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q, d, k = 100_000, 1024, 100
eng = KnnEngine(0, profile=True)
eng.set_prescreen(bool(pre))
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
for _ in range(2):
    eng.search(q, k, "sqeuclidean")
    s = eng.stats()
    print({key: (round(v, 2) if isinstance(v, float) else v) for key, v in s.items()}, flush=True)
