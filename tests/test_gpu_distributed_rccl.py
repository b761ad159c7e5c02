"""GPU: the production backend of the one-process-per-GPU exchange is nccl == RCCL.  The box has one GPU, so the group has one rank -- but
init_process_group("nccl"), all_gather_into_tensor into the stacked buffer, the HIP-event timing of the collective and the merge kernel
behind it all run for real (round 2's verdict: "the nccl branch has never executed anywhere")."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

SCRIPT = r'''
import os, sys, json
sys.path.insert(0, os.environ["NW_ROOT"])
import torch, torch.distributed as td
from neighborhoodwatch_amd import KnnEngine
from neighborhoodwatch_amd.distributed import allgather_partials, _timed_allgather, shard_bounds
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = KnnEngine(0)
base = eng.generate(0x6E770001, 0, 50_000, 96); queries = eng.generate(0x6E770002, 0, 300, 96)
eng.set_base(base)
d, i = eng.search(queries, 20, index_offset=shard_bounds(50_000, 1, 0)[0])
timings = {}
gd, gi = _timed_allgather(d, i, None, timings)                   # RCCL all-gather of the [Q,k] partials (world of one: a copy through RCCL)
assert td.get_backend() == "nccl" and gd.shape == (1, 300, 20) and gi.shape == (1, 300, 20)
assert torch.equal(gd[0], d) and torch.equal(gi[0], i)
md, mi = eng.merge(torch.cat([gd, gd]), torch.cat([gi, gi + 1_000_000]), 20)      # two "ranks": the same lists with shifted ids
assert torch.equal(mi[:, 0], i[:, 0]) and torch.equal(md[:, 0], d[:, 0])
t = torch.ones(4, device="cuda"); td.all_reduce(t); assert float(t.sum()) == 4.0
print(json.dumps({"ok": True, "allgather_ms": timings["allgather_ms"], "bytes": timings["allgather_bytes"], "rccl": torch.cuda.nccl.version()}))
td.barrier(); td.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_allgather_branch_runs_on_the_gpu_box(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NW_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NWKNN_DIST_BACKEND"):
        env.pop(var, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    import json
    line = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][-1]
    assert line["ok"] is True and line["allgather_ms"] > 0.0 and line["bytes"] == 300 * 20 * 12


@pytest.mark.gpu
@pytest.mark.parametrize("n_lists,k,n_slices", [(8, 1000, 3), (5, 100, 4), (64, 64, 7), (2, 10, 1), (3, 33, 50)])
def test_the_sliced_and_two_level_merge_of_a_multi_device_search(oracle, n_lists, k, n_slices):
    """What devices 1..N-1 of nwknn_search_sharded do has never run on hardware (one-GPU boxes): a query slice that does not start at row
    0, lists whose stride is not the slice's own size, and -- from 4096 (list, rank) entries per query on -- the two-level fold.  The same
    code (merge_slice behind nwknn_comm_merge_lists) is driven here on one device with caller-built lists: random partial lists with
    cross-list equal distances (ties go to the lower index), empty slots, more slices than rows in one case; checked against the oracle's
    merge (= the reference's fold, tests/golden/merge_*.npz) bit for bit."""
    import numpy as np
    from neighborhoodwatch_amd import ShardedKnnEngine
    rng = np.random.default_rng(n_lists * 1000 + k)
    nq = 37
    pool = np.sort(rng.random(200).astype(np.float32))                      # few distinct distances: ties within and across lists
    dists = np.sort(rng.choice(pool, size=(n_lists, nq, k)), axis=2).astype(np.float32)
    idx = np.empty((n_lists, nq, k), dtype=np.int64)
    for l in range(n_lists):                                                 # shard l owns ids [l * 10^6, (l+1) * 10^6): distinct within a query
        for q in range(nq):
            ids = np.sort(rng.choice(1_000_000, size=k, replace=False)) + l * 1_000_000
            order = np.lexsort((ids, dists[l, q]))                           # a partial list is sorted by (distance, index)
            idx[l, q] = ids[order]
            dists[l, q] = dists[l, q][order]
    idx[-1, :, k - k // 4:] = -1                                             # the last list is short: empty slots
    dists[-1, :, k - k // 4:] = np.inf
    rd, ri = oracle.merge_partials(dists, idx, k)
    with ShardedKnnEngine([0]) as eng:
        od, oi = eng.merge_lists(dists, idx, k, n_slices)
    assert np.array_equal(oi, ri)
    assert np.array_equal(od.view(np.uint32), rd.view(np.uint32))
