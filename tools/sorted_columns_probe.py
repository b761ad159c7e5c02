"""Probe: how much of the pre-screen epilogue's cost is false positives of the first-pass block test that come from the COLUMN side
(the lane's four columns c, c+16, c+32, c+48 of a wave's 64-column range share one threshold, the smallest of their four column terms)?
The same search on a base whose rows are re-arranged inside every 64-row group so that those four columns hold neighbours in norm order
(and, optionally, queries re-arranged likewise inside every 16-row group): tile TF before / after.
usage: sorted_columns_probe.py [d] [B] [Q] [rows 0|1]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 32_768
rows_too = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = KnnEngine(0, profile=True)
base = eng.generate(0x6E770001, 0, B, d)
queries = eng.generate(0x6E770002, 0, Q, d)


def deal(x, group, lanes, key=None):
    """inside every `group` rows: sort by squared norm (or `key`), rank r goes to position (r // (group // lanes)) + lanes * (r % (group // lanes))"""
    n = x.shape[0] // group * group
    nrm = ((x[:n].float() ** 2).sum(1) if key is None else key[:n]).view(-1, group)
    order = nrm.argsort(dim=1)                                   # order[g, r] = local index of the row of rank r
    per = group // lanes
    r = torch.arange(group, device=x.device)
    pos = (r // per) + lanes * (r % per)
    src = torch.empty_like(order)
    src[:, pos] = order                                          # position pos(r) takes the row of rank r
    idx = (src + torch.arange(0, n, group, device=x.device).view(-1, 1)).view(-1)
    out = x.clone()
    out[:n] = x[idx]
    return out


CENSUS = "dev" in __import__("os").environ.get("NWKNN_LIB_PATH", "")     # developer build: also count the flagged blocks (tools/f16x_census.py)


def run(tag, b, q):
    eng.set_prescreen(True)
    eng.set_base(b)
    for _ in range(3):
        eng.search(q, 100)
        s = eng.stats()
    if CENSUS:
        dbg = torch.zeros(((1 << 23) + 64,), dtype=torch.int64, device="cuda")
        eng.set_option("dbg_ptr", dbg.data_ptr()); eng.set_option("diag", 64)
        eng.search(q, 100)
        torch.cuda.synchronize()
        c = dbg[(1 << 23):(1 << 23) + 4].cpu().numpy()
        eng.set_option("diag", 0); eng.set_option("dbg_ptr", 0)
        print(f"   wave-tiles {c[0]}: block mask fires in {c[1] / c[0]:.1%}, flagged blocks per wave-tile {c[3] / c[0]:.3f}")
    print(f"{tag}: tile {s['tile_flops'] / 1e12 / (s['tile_ms'] / 1e3):.0f} TF, search {s['total_ms']:.1f} ms, rescore {s['rescore_ms']:.1f} ms", flush=True)


run("as generated", base, queries)
b2 = deal(base, 64, 16)
run("columns dealt by norm inside 64-row groups", b2, queries)
if rows_too:
    q2 = deal(queries, 16, 4)
    run("+ query rows dealt by norm inside 16-row groups", b2, q2)
    dist, _ = eng.search(queries, 100)
    key = (queries.float() ** 2).sum(1) - torch.as_tensor(dist)[:, 99].to(queries.device)      # the row term of squared L2: |q|^2 - tau
    q3 = deal(queries, 16, 4, key)
    run("+ query rows dealt by (|q|^2 - final tau) inside 16-row groups", b2, q3)
    q4 = queries[key.argsort()]
    run("+ all query rows sorted by (|q|^2 - final tau)", b2, q4)
