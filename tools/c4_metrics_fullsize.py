"""One-off full-size parity of the C4 per-GPU share (1M queries x 12.5M fp16 rows, d = 1536) under the two metrics the GPU suite does not run at
that size: one_minus_dot (engine='torch') and cosine (engine='cuvs').  Properties on every result row + sampled rows recomputed by the CPU oracle
against the ENTIRE shard, bit for bit.  usage: c4_metrics_fullsize.py [queries]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from neighborhoodwatch_amd import KnnEngine
from oracle import cpu_oracle as orc                      # the checker
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B, d, k, offset = 12_500_000, 1536, 100, 37_500_000
rows = np.linspace(0, Q - 1, 8).astype(np.int64)
with KnnEngine(0, profile=True) as eng:
    base = eng.generate(orc.SEED_BASE, offset, B, d, "f16")
    queries = eng.generate(orc.SEED_QUERY, 0, Q, d, "f16")
    eng.set_base(base)
    got = {}
    for metric in ("one_minus_dot", "cosine"):
        t0 = time.perf_counter()
        dist, idx = eng.search(queries, k, metric, index_offset=offset)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = eng.stats()
        assert st["prescreen"] == 1 and st["overflowed_queries"] == 0 and st["prescreen_violations"] == 0 and st["prescreen_fallbacks"] == 0, st
        assert bool((dist[:, 1:] >= dist[:, :-1]).all()) and int(idx.min()) >= offset and int(idx.max()) < offset + B
        sel = torch.from_numpy(rows).cuda()
        got[metric] = (dist[sel].cpu().numpy(), idx[sel].cpu().numpy())
        print(f"{metric}: {Q} x {B} in {dt:.2f} s = {Q / dt:.0f} q/s, tile {st['tile_flops'] / st['tile_ms'] / 1e9:.1f} TF, rescore {st['rescore_ms']:.0f} ms, "
              f"invariant {st['prescreen_checked']} pairs, worst {st['prescreen_worst']:.4f}, kernel {eng.tile_kernel_name()}", flush=True)
        del dist, idx
    hq = queries[torch.from_numpy(rows).cuda()].float().cpu().numpy()
    del queries
    hb = np.empty((B, d), dtype=np.float32)
    for s in range(0, B, 1_250_000):
        hb[s:s + 1_250_000] = base[s:s + 1_250_000].float().cpu().numpy()
    del base
for metric, (gd, gi) in got.items():
    rd, ri = orc.knn_canonical(hb, hq, k, metric)
    ok = np.array_equal(gi, ri + offset) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32))
    print(f"{metric}: {len(rows)} sampled rows vs the oracle over the whole shard: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
