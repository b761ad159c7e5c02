"""CPU, world_size 2, gloo: the multi-rank path (shard -> search with global ids -> all-gather -> merge).
The GPU kernels are replaced by the oracle HERE ONLY (tests may use it); production passes KnnEngine.search/merge."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as td
    from neighborhoodwatch_amd.distributed import shard_bounds, sharded_search
    from oracle import cpu_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    B, Q, d, k = 1003, 21, 40, 16
    lo, hi = shard_bounds(B, world, rank)
    base = orc.gen(orc.SEED_BASE, lo, hi - lo, d)          # every rank generates only its shard
    queries = torch.from_numpy(orc.gen(orc.SEED_QUERY, 0, Q, d))

    def search_shard(qs, kk, offset):
        dd, ii = orc.knn_canonical(base, qs.numpy(), kk, "sqeuclidean")
        return torch.from_numpy(dd), torch.from_numpy(ii + offset)

    merged_rows = []

    def merge(gd, gi, kk):
        assert gd.shape[0] == world and gd.is_contiguous() and gi.is_contiguous()
        merged_rows.append(gd.shape[1])
        dd, ii = orc.merge_partials(gd.numpy(), gi.numpy(), kk)
        return torch.from_numpy(dd), torch.from_numpy(ii)

    timings = {}
    dist, idx = sharded_search(search_shard, merge, queries, k, lo, None, timings)               # default: rank g merges its slice of the query rows
    qlo, qhi = shard_bounds(Q, world, rank)
    assert merged_rows == [qhi - qlo], merged_rows                                                 # 21 queries over 2 ranks: 10 and 11 rows, not 21
    assert timings["allgather_bytes"] == world * Q * k * 12 and timings["slices_bytes"] == world * 11 * k * 12 and timings["merge_ms"] >= 0.0
    rdist, ridx = sharded_search(search_shard, merge, queries, k, lo, None, None, "replicated")    # every rank merges everything: the same table
    assert merged_rows == [qhi - qlo, Q] and torch.equal(rdist, dist) and torch.equal(ridx, idx)
    np.savez(Path(out_dir) / f"rank{rank}.npz", dist=dist.numpy(), idx=idx.numpy())
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharded_search_equals_single_shard(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    base = oracle.gen(oracle.SEED_BASE, 0, 1003, 40)
    queries = oracle.gen(oracle.SEED_QUERY, 0, 21, 40)
    rd, ri = oracle.knn_canonical(base, queries, 16, "sqeuclidean")
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["idx"], ri), f"rank {r}: merged ids differ from the single-shard result"
        assert np.array_equal(g["dist"].view(np.uint32), rd.view(np.uint32))


def _worker_q(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as td
    from neighborhoodwatch_amd.distributed import query_sharded_search
    from oracle import cpu_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    B, Q, d, k = 700, 23, 24, 9                                   # 23 queries over 2 ranks: slices of 11 and 12 rows
    base = orc.gen(orc.SEED_BASE, 0, B, d)                        # every rank holds the whole base
    queries = torch.from_numpy(orc.gen(orc.SEED_QUERY, 0, Q, d))

    def search_all(qs, kk):
        dd, ii = orc.knn_canonical(base, qs.numpy(), kk, "sqeuclidean")
        return torch.from_numpy(dd), torch.from_numpy(ii)

    dist, idx = query_sharded_search(search_all, queries, k, None)
    np.savez(Path(out_dir) / f"qrank{rank}.npz", dist=dist.numpy(), idx=idx.numpy())
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_query_sharded_search_equals_single_rank(tmp_path, oracle):
    world = 2
    mp.spawn(_worker_q, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    base = oracle.gen(oracle.SEED_BASE, 0, 700, 24)
    queries = oracle.gen(oracle.SEED_QUERY, 0, 23, 24)
    rd, ri = oracle.knn_canonical(base, queries, 9, "sqeuclidean")
    for r in range(world):
        g = np.load(tmp_path / f"qrank{r}.npz")
        assert np.array_equal(g["idx"], ri) and np.array_equal(g["dist"].view(np.uint32), rd.view(np.uint32))


def _worker_ties(rank, world, port, out_dir):
    """Three uneven shards (1003 rows: 334/334/335) whose rows repeat across shard borders: a query's equal-distance neighbours
    live on different ranks, so the merged order is decided by the GLOBAL row id (lower wins), not by the rank that sent it."""
    sys.path.insert(0, str(ROOT))
    import torch.distributed as td
    from neighborhoodwatch_amd.distributed import shard_bounds, sharded_search
    from oracle import cpu_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    B, Q, d, k = 1003, 17, 32, 24
    lo, hi = shard_bounds(B, world, rank)
    base = _tied_base(orc, B, d)[lo:hi]
    queries = torch.from_numpy(orc.gen(orc.SEED_QUERY, 0, Q, d))

    def search_shard(qs, kk, offset):
        dd, ii = orc.knn_canonical(base, qs.numpy(), kk, "one_minus_dot")
        return torch.from_numpy(dd), torch.from_numpy(ii + offset)

    def merge(gd, gi, kk):
        dd, ii = orc.merge_partials(gd.numpy(), gi.numpy(), kk)
        return torch.from_numpy(dd), torch.from_numpy(ii)

    dist, idx = sharded_search(search_shard, merge, queries, k, lo, None)
    np.savez(Path(out_dir) / f"trank{rank}.npz", dist=dist.numpy(), idx=idx.numpy())
    td.barrier()
    td.destroy_process_group()


def _tied_base(orc, B, d):
    """Only 97 distinct rows: row r repeats row r % 97, so every neighbour has ~10 exact copies spread over all shards."""
    distinct = orc.gen(orc.SEED_BASE, 0, 97, d)
    return np.ascontiguousarray(distinct[np.arange(B) % 97])


@pytest.mark.timeout(180)
def test_three_uneven_ranks_break_cross_shard_ties_by_global_row_id(tmp_path, oracle):
    world = 3
    mp.spawn(_worker_ties, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    base = _tied_base(oracle, 1003, 32)
    queries = oracle.gen(oracle.SEED_QUERY, 0, 17, 32)
    rd, ri = oracle.knn_canonical(base, queries, 24, "one_minus_dot")
    assert (np.diff(rd, axis=1) == 0).any()                        # the case really holds ties
    tied = np.diff(rd, axis=1) == 0
    assert (np.diff(ri, axis=1)[tied] > 0).all()                   # ... which the single-shard result orders by row id
    for r in range(world):
        g = np.load(tmp_path / f"trank{r}.npz")
        assert np.array_equal(g["idx"], ri), f"rank {r}: merged ids differ from the single-shard result"
        assert np.array_equal(g["dist"].view(np.uint32), rd.view(np.uint32))


def _worker_few(rank, world, port, out_dir):
    """Fewer queries than ranks (2 over 3): the rank with an empty slice contributes padding only; nobody hangs in the all-gather."""
    sys.path.insert(0, str(ROOT))
    import torch.distributed as td
    from neighborhoodwatch_amd.distributed import query_sharded_search
    from oracle import cpu_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    B, Q, d, k = 300, 2, 16, 5
    base = orc.gen(orc.SEED_BASE, 0, B, d)
    queries = torch.from_numpy(orc.gen(orc.SEED_QUERY, 0, Q, d))
    calls = []

    def search_all(qs, kk):
        assert qs.shape[0] > 0, "an empty slice must not reach the engine"
        calls.append(qs.shape[0])
        dd, ii = orc.knn_canonical(base, qs.numpy(), kk, "sqeuclidean")
        return torch.from_numpy(dd), torch.from_numpy(ii)

    timings = {}
    dist, idx = query_sharded_search(search_all, queries, k, None, timings)
    assert timings["allgather_ms"] >= 0.0 and timings["allgather_bytes"] == world * 1 * k * 12
    np.savez(Path(out_dir) / f"frank{rank}.npz", dist=dist.numpy(), idx=idx.numpy(), calls=np.array(calls, dtype=np.int64))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.timeout(180)
def test_fewer_queries_than_ranks_does_not_hang(tmp_path, oracle):
    world = 3
    mp.spawn(_worker_few, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    base = oracle.gen(oracle.SEED_BASE, 0, 300, 16)
    queries = oracle.gen(oracle.SEED_QUERY, 0, 2, 16)
    rd, ri = oracle.knn_canonical(base, queries, 5, "sqeuclidean")
    n_calls = 0
    for r in range(world):
        g = np.load(tmp_path / f"frank{r}.npz")
        assert np.array_equal(g["idx"], ri) and np.array_equal(g["dist"].view(np.uint32), rd.view(np.uint32))
        n_calls += len(g["calls"])
    assert n_calls == 2                                           # shard_bounds(2, 3, r): one rank holds no query row
