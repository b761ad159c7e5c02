"""Per-launch time of the tile kernel over the chunk walk of one search (run under rocprofv3 --kernel-trace), for one 8-GPU
shard of C3: 100k queries x 1.25M base rows, d = 1024, pre-screened.  usage: chunk_times.py [B] [prescreen 0|1] [Q] [d] [opts k=v,...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
k = 100
eng = KnnEngine(0, profile=True)
eng.set_prescreen(bool(pre))
for kv in (sys.argv[5].split(',') if len(sys.argv) > 5 else []):
    eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
eng.set_base(base)
for _ in range(2):
    eng.search(q, k, "sqeuclidean")
    s = eng.stats()
    print({key: (round(v, 2) if isinstance(v, float) else v) for key, v in s.items()}, flush=True)
