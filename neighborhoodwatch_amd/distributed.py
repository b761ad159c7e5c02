"""Base-sharded multi-GPU search: one process per GPU, one collective.

The reference is single-GPU (device 0 hard-coded, cu_knn.py:65) and folds per-base-tile partial tables on the
CPU (merge.py:56-101).  Here the base rows are split contiguously over the ranks of a ``torch.distributed``
group (backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests), every rank searches its own shard
with global row ids (``index_offset`` = first row of the shard, the analogue of ``+ batch_offset``,
cu_knn.py:273), then ONE all-gather moves the [Q,k] partial lists and a merge by (distance, index) gives the
final table on every rank.  Payload: Q*k*12 bytes per rank (C3: 120 MB) -- negligible next to the search.

The search and merge callables are injected so the same code path runs under gloo on CPU in tests (with the
oracle standing in for the GPU kernels -- test only) and under RCCL in production (KnnEngine).
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of shard `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def allgather_partials(dist, idx, group=None):
    """[Q,k] per rank -> ([world,Q,k] distances, [world,Q,k] indices) on every rank, rank order == row order."""
    import torch
    import torch.distributed as td
    world = td.get_world_size(group)
    gd = torch.empty((world,) + tuple(dist.shape), dtype=dist.dtype, device=dist.device)
    gi = torch.empty((world,) + tuple(idx.shape), dtype=idx.dtype, device=idx.device)
    if td.get_backend(group) == "nccl":              # RCCL: one ncclAllGather per array straight into the stacked buffer
        td.all_gather_into_tensor(gd, dist.contiguous(), group=group)
        td.all_gather_into_tensor(gi, idx.contiguous(), group=group)
        return gd, gi
    # gloo (CPU tests, and single-GPU rehearsals of the N-rank path): list form; device tensors are staged through the host
    hd, hi = dist.contiguous().cpu(), idx.contiguous().cpu()
    cd = torch.empty((world,) + tuple(hd.shape), dtype=hd.dtype)
    ci = torch.empty((world,) + tuple(hi.shape), dtype=hi.dtype)
    td.all_gather(list(cd.unbind(0)), hd, group=group)
    td.all_gather(list(ci.unbind(0)), hi, group=group)
    return cd.to(dist.device), ci.to(idx.device)


def _timed_allgather(dist, idx, group, timings):
    """allgather_partials with its duration added to ``timings['allgather_ms']``: HIP events on torch's current stream for
    device tensors under RCCL (the collective runs on RCCL's stream, which the current stream waits for), the host clock otherwise."""
    import time
    import torch
    import torch.distributed as td
    if timings is None:
        return allgather_partials(dist, idx, group)
    on_dev = dist.is_cuda and td.get_backend(group) == "nccl"
    if on_dev:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = allgather_partials(dist, idx, group)
        e1.record()
        e1.synchronize()
        timings["allgather_ms"] = timings.get("allgather_ms", 0.0) + e0.elapsed_time(e1)
    else:
        t0 = time.perf_counter()
        out = allgather_partials(dist, idx, group)
        if dist.is_cuda:
            torch.cuda.synchronize(dist.device)
        timings["allgather_ms"] = timings.get("allgather_ms", 0.0) + (time.perf_counter() - t0) * 1e3
    timings["allgather_bytes"] = int(out[0].numel() * 4 + out[1].numel() * 8)
    return out


def _allgather_row_slices(part_d, part_i, nq: int, k: int, like, group, timings, key: str):
    """Rank r holds rows shard_bounds(nq, world, r) of a [nq,k] table (``part_*`` = its slice, or None when the slice is empty): pad the
    slices to the longest, ONE all-gather per array, concatenate in rank order -> the whole table on every rank."""
    import torch
    import torch.distributed as td
    world, rank = td.get_world_size(group), td.get_rank(group)
    lo, hi = shard_bounds(nq, world, rank)
    rows = -(-nq // world)                                        # slices differ by at most one row: pad to the longest
    pd_ = torch.zeros((rows, k), dtype=torch.float32, device=like.device)
    pi_ = torch.zeros((rows, k), dtype=torch.int64, device=like.device)
    if hi > lo:
        pd_[:hi - lo] = part_d
        pi_[:hi - lo] = part_i
    sub = {} if timings is not None else None
    gd, gi = _timed_allgather(pd_, pi_, group, sub)               # [world, rows, k]
    if timings is not None:
        timings[key] = timings.get(key, 0.0) + sub["allgather_ms"]
        timings[key.replace("_ms", "_bytes")] = sub["allgather_bytes"]
    parts = [shard_bounds(nq, world, r) for r in range(world)]
    return (torch.cat([gd[r, :b - a] for r, (a, b) in enumerate(parts)]), torch.cat([gi[r, :b - a] for r, (a, b) in enumerate(parts)]))


def sharded_search(search_shard: Callable, merge: Callable, queries, k: int, shard_lo: int, group=None, timings=None, merge_mode: str = "slice"):
    """search_shard(queries, k, index_offset) -> (dist [Q,k] f32 tensor, idx [Q,k] i64 tensor, global ids);
    merge(dists [W,q,k], idx [W,q,k], k) -> ([q,k], [q,k]).  Returns the merged table (identical on all ranks).

    After the all-gather of the [Q,k] partial lists every rank holds all W lists.  ``merge_mode="slice"`` (default; what the C ABI's
    nwknn_search_sharded does, nwknn_comm.cpp): rank g merges only ITS slice of the query rows (shard_bounds over the queries) -- 1/W of
    the merge work -- and a second, small all-gather (Q*k*12 bytes in total) hands every rank the whole table.  ``"replicated"``: every
    rank merges all Q rows (the first two rounds' behaviour; no second exchange).  Same result, bit for bit.
    ``timings`` (a dict) accumulates ``allgather_ms`` / ``merge_ms`` (/ ``slices_ms``: the second all-gather) of the exchange step (the
    merge call is blocking: host clock)."""
    import time
    import torch.distributed as td
    if merge_mode not in ("slice", "replicated"):
        raise ValueError(f"merge_mode must be 'slice' or 'replicated', got {merge_mode!r}")
    dist, idx = search_shard(queries, k, shard_lo)
    if not td.is_initialized() or td.get_world_size(group) == 1:
        return dist, idx
    gd, gi = _timed_allgather(dist, idx, group, timings)

    def timed_merge(d_, i_):
        t0 = time.perf_counter()
        out = merge(d_, i_, k)
        if timings is not None:
            if getattr(out[0], "is_cuda", False):
                import torch
                torch.cuda.synchronize(out[0].device)
            timings["merge_ms"] = timings.get("merge_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    if merge_mode == "replicated":
        return timed_merge(gd, gi)
    nq = dist.shape[0]
    lo, hi = shard_bounds(nq, td.get_world_size(group), td.get_rank(group))
    md, mi = timed_merge(gd[:, lo:hi].contiguous(), gi[:, lo:hi].contiguous()) if hi > lo else (None, None)
    return _allgather_row_slices(md, mi, nq, k, dist, group, timings, "slices_ms")


def query_sharded_search(search_all: Callable, queries, k: int, group=None, timings=None):
    """The zero-collective alternative for jobs whose base fits one GPU (C2, C3): every rank holds the WHOLE base and
    searches its contiguous slice of the query rows (``shard_bounds`` over the queries); the slices are concatenated in rank
    order by one all-gather -- there is nothing to merge, a query's neighbours all come from one rank.
    search_all(queries_slice, k) -> (dist [q,k] f32 tensor, idx [q,k] i64 tensor).  Returns ([Q,k], [Q,k]) on every rank."""
    import torch.distributed as td
    if not td.is_initialized() or td.get_world_size(group) == 1:
        return search_all(queries, k)
    world, rank = td.get_world_size(group), td.get_rank(group)
    nq = queries.shape[0]
    lo, hi = shard_bounds(nq, world, rank)
    # fewer queries than ranks: a rank with an empty slice searches nothing and contributes only padding (the engine refuses an
    # empty query set, and a rank that raised here would leave the others blocked in the all-gather)
    dist, idx = search_all(queries[lo:hi], k) if hi > lo else (None, None)
    return _allgather_row_slices(dist, idx, nq, k, queries, group, timings, "allgather_ms")
