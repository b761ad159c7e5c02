import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 2048, 131072, 768, 100
for metric in ("sqeuclidean", "one_minus_dot", "cosine"):
    for opts in ({}, {"prescreen_prune": 0}, {"pre_eps_pct": 0}):
        eng = KnnEngine(0, profile=True)
        eng.set_prescreen(True)
        for kk, v in opts.items(): eng.set_option(kk, v)
        base = eng.generate(1, 0, B, d); q = eng.generate(2, 0, Q, d)
        eng.set_base(base)
        eng.search(q, k, metric)
        s = eng.stats()
        print(metric, opts, {x: (round(s[x], 2) if isinstance(s[x], float) else s[x]) for x in ("prescreen", "tile_launches", "rescan_launches", "overflowed_queries", "tile_ms", "rescore_ms")}, flush=True)
        eng.close()
