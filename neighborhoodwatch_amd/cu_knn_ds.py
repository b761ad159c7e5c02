"""kNN driver, pyarrow.dataset flavour -- same entry point as the reference's ``neighborhoodwatch/cu_knn_ds.py``.

    compute_knn_ds(data_dir, dimensions, query_filename, query_count, base_filename, base_count,
                   mem_tune=False, k=100, initial_batch_size=200000, max_memory_threshold=0.2, split=True)

The base file is STREAMED in batches of ``batch_size`` rows (cu_knn_ds.py:181-239): a reader thread decodes batch s+1 from its own
parquet row groups while batch s is on the GPU (host memory holds at most two batches, never the file); each batch is searched
with global row ids (``+ batch_offset``, cu_knn_ds.py:216) and folded into the running ``[Q,k]`` result by the device merge kernel --
the running pair lives in HBM between batches, nothing round-trips through numpy -- and ``partial/final_distances.parquet`` /
``final_indices.parquet`` are written ONCE at the end.  (The reference writes the final files inside the loop, cu_knn_ds.py:232-233,
so every base batch overwrites the previous one and the result is only right for a single batch; merge then finds no partial files
and is a no-op, merge.py:40-42.)  ``resume=True`` checkpoints the running pair after every batch (written behind the GPU's back by
a writer thread) and continues an interrupted run from the last checkpoint whose metric / k / row counts match.
"""
from __future__ import annotations

import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pyarrow.dataset as ds
import pyarrow.parquet as pq

from .cu_knn import (PARQUET_ROW_GROUP, cleanup, engine_search, engine_set_base, float_row_bytes, list_embedding_field, stream_cudf_to_parquet,
                     stream_tiles, table_columns, table_input)
from .engine import default_engine, mem_info, resolve_metric
from .nw_utils import get_full_filename, get_partial_distances_filename, get_partial_indices_filename
from .parquet_stream import ParquetRows, Prefetch, TableRows


PREFETCH_SLOTS = 2                  # base batches alive on the host at once: the one on the GPU + the one being decoded (1 = no overlap)


def get_gpu_memory(device: int = 0):
    """Total HBM bytes of one GPU (cu_knn_ds.py:30-36 asks NVML for card 0)."""
    return mem_info(device)[1]


def load_dataset(data_dir, filename):
    return ds.dataset(get_full_filename(data_dir, filename), format="parquet")


def dataset_count(dataset):
    if isinstance(dataset, ParquetRows):
        return len(dataset)
    return dataset.count_rows()                                   # parquet metadata only


def slice_dataset(dataset, start, end):
    """Rows [start, end) of a dataset (cu_knn_ds.py:48-49).  The reference materialises them (``dataset.take(arange)``); for the
    only use it makes of this -- ``slice_dataset(ds, 0, count)`` over parquet files -- a lazy ``ParquetRows`` window is returned
    instead: nothing is read until a batch is asked for.  Other cases fall back to the reference's materialising form."""
    files = getattr(dataset, "files", None)
    if start == 0 and files:
        return ParquetRows(files, end)
    return ds.dataset(dataset.take(np.arange(start, min(end, dataset_count(dataset)))))


def get_dataset_columns(dataset, n):
    """Columns handed to the scanner: the text column if present plus embedding_0..n-1 (cu_knn_ds.py:122-133);
    only the float columns reach the GPU."""
    names = dataset.schema.names
    cols = [c for c in ("text", "question") if c in names][-1:]
    lst = list_embedding_field(dataset.schema)
    if lst is not None:                                           # the one-column FixedSizeList layout
        assert lst.type.list_size == n, f"Expected {n} values, got {lst.type.list_size}"
        return cols + [lst.name]
    return cols + [f"embedding_{i}" for i in range(n)]


def tune_memory(dataset, batch_size, max_memory_threshold, rmm, column_names, device: int = 0):
    """cu_knn_ds.py:57-109 as arithmetic (see cu_knn.tune_memory): here the first jump already uses
    ``max_memory_threshold`` (cu_knn_ds.py:69), so the loop's first probe sits at the threshold."""
    num_rows = dataset_count(dataset)
    total = get_gpu_memory(device)
    print("-- total memory per GPU:", total)
    schema = dataset.schema
    have = set(schema.names)
    fields = [schema.field(c) for c in column_names if c in have]
    row_bytes = float_row_bytes(fields)
    used = min(batch_size, num_rows) * row_bytes
    factor = math.ceil((total * max_memory_threshold) / used)
    print(f"-- batch processing adjustment factor (per GPU): {factor}")
    batch_size *= factor
    if num_rows < batch_size:
        return num_rows
    if batch_size * row_bytes > max_memory_threshold * total:
        return int(0.8 * batch_size)
    grown = batch_size * 1.2
    return num_rows if num_rows < grown else int(0.8 * grown)


def compute_knn_ds(data_dir,
                   dimensions,
                   query_filename,
                   query_count,
                   base_filename,
                   base_count,
                   mem_tune=False,
                   k=100,
                   initial_batch_size=200000,
                   max_memory_threshold=0.2,
                   split=True,
                   *,
                   engine='raft',
                   metric=None,
                   device: int = 0,
                   prescreen: bool = False,
                   resume: bool = False):
    batch_size = initial_batch_size
    query_dataset = slice_dataset(load_dataset(data_dir, query_filename), 0, query_count)
    query_column_names = get_dataset_columns(query_dataset, dimensions)
    base_dataset = slice_dataset(load_dataset(data_dir, base_filename), 0, base_count)
    base_column_names = get_dataset_columns(base_dataset, dimensions)
    if mem_tune:
        batch_size = tune_memory(base_dataset, batch_size, max_memory_threshold, None, base_column_names, device)
    assert (base_count % batch_size == 0) or k <= (base_count % batch_size), \
        f"Cannot generate k of {k} with only {base_count} rows and batch_size of {batch_size}."           # cu_knn_ds.py:167
    process_dataset_batches(data_dir, base_dataset, base_column_names, query_dataset, query_column_names, batch_size,
                            query_count, k, split, engine=engine, metric=metric, device=device, prescreen=prescreen, resume=resume)


def _embedding_columns(names):
    return [c for c in names if c.startswith("embedding_") or c == "embedding"]


def _as_rows(dataset, column_names) -> ParquetRows:
    """A dataset argument as a lazy row window over its float columns (the text column the reference's scanner also carries,
    cu_knn_ds.py:122-133, never reaches the GPU and is not decoded here)."""
    cols = _embedding_columns(column_names)
    if isinstance(dataset, (ParquetRows, TableRows)):
        return dataset.select(cols)
    files = getattr(dataset, "files", None)
    if files:
        return ParquetRows(files, None, cols)
    if isinstance(dataset, pa.Table):
        return TableRows(dataset.select(cols))
    if hasattr(dataset, "to_table"):                              # an in-memory pyarrow dataset (slice_dataset's materialising fallback): no files behind it
        return TableRows(dataset.to_table(columns=cols))
    raise TypeError(f"cannot read rows from a {type(dataset).__name__}: expected a parquet-backed or in-memory pyarrow dataset, a Table or ParquetRows")


# ---- resume: the running [Q,k] pair after `rows_done` base rows ------------------------------------------------------------
def _checkpoint_names(data_dir):
    return f"{data_dir}/partial/running_distances.parquet", f"{data_dir}/partial/running_indices.parquet"


def _checkpoint_tag(metric, k, query_count, base_rows, rows_done, inputs: str = "") -> dict:
    """What a checkpoint was computed from.  ``inputs`` ties it to the data: the provenance of the query and base rows (file paths, sizes,
    modification times, columns) -- equal counts alone would let a stale checkpoint of ANOTHER job in the same data_dir be resumed."""
    return {"nwknn.metric": metric, "nwknn.k": str(k), "nwknn.queries": str(query_count), "nwknn.base_rows": str(base_rows),
            "nwknn.rows_done": str(rows_done), "nwknn.inputs": inputs}


def checkpoint_inputs(query_rows, base_rows) -> str:
    """The identity of a job's input rows as a checkpoint records it (``ParquetRows.provenance``: file paths, sizes, mtimes, row window,
    columns)."""
    return f"q:{query_rows.provenance()};b:{base_rows.provenance()}"


def _write_checkpoint(data_dir, dist, idx, tag):
    fd, fi = _checkpoint_names(data_dir)
    for name, arr in ((fd, dist), (fi, idx)):                    # whole files, swapped in by rename: a reader never sees half a table
        stream_cudf_to_parquet(name + ".tmp", arr, PARQUET_ROW_GROUP, tag)
        os.replace(name + ".tmp", name)


def _load_checkpoint(data_dir, metric, k, query_count, base_rows, inputs: str = ""):
    """-> (rows_done, distances [Q,k] f32, indices [Q,k] i64) of a checkpoint written by a run of the same job, or None."""
    fd, fi = _checkpoint_names(data_dir)
    try:
        td, ti = pq.read_table(fd), pq.read_table(fi)
        kd, ki = td.schema.metadata or {}, ti.schema.metadata or {}
        rows_done = int(kd[b"nwknn.rows_done"])
        want = _checkpoint_tag(metric, k, query_count, base_rows, rows_done, inputs)
        if any(kd.get(a.encode()) != b.encode() or ki.get(a.encode()) != b.encode() for a, b in want.items()):
            return None
        if len(td) != query_count or len(ti) != query_count or td.num_columns != k or ti.num_columns != k or not (0 < rows_done <= base_rows):
            return None
        dist = np.stack([td.column(str(j)).to_numpy() for j in range(k)], axis=1).astype(np.float32)
        idx = np.stack([ti.column(str(j)).to_numpy() for j in range(k)], axis=1).astype(np.int64)
        return rows_done, np.ascontiguousarray(dist), np.ascontiguousarray(idx)
    except (OSError, KeyError, ValueError, pa.ArrowInvalid):
        return None


def _drop_checkpoint(data_dir):
    for name in _checkpoint_names(data_dir):
        for f in (name, name + ".tmp"):
            try:
                os.remove(f)
            except OSError:
                pass


def process_dataset_batches(data_dir,
                            base_dataset,
                            base_column_names,
                            query_dataset,
                            query_column_names,
                            batch_size,
                            query_count,
                            k,
                            split,
                            *,
                            engine='raft',
                            metric=None,
                            device: int = 0,
                            prescreen: bool = False,
                            resume: bool = False):
    import torch
    metric = resolve_metric(metric, engine)
    eng = default_engine(device)
    eng.set_prescreen(prescreen)
    query_rows = _as_rows(query_dataset, query_column_names)
    query_table = query_rows.to_table()
    queries = table_input(query_table)
    assert len(query_table) == query_count                                        # cu_knn_ds.py:229-230
    base_rows = _as_rows(base_dataset, base_column_names)
    n_base = len(base_rows)
    os.makedirs(f"{data_dir}/partial", exist_ok=True)

    run_d = run_i = None                                                          # the running pair: torch CUDA tensors [Q,k]
    start_row = 0
    inputs = checkpoint_inputs(query_rows, base_rows)
    ckpt = _load_checkpoint(data_dir, metric, k, query_count, n_base, inputs) if resume else None
    if ckpt is not None:
        start_row = ckpt[0]
        dev = torch.device("cuda", device)
        run_d, run_i = torch.from_numpy(ckpt[1]).to(dev), torch.from_numpy(ckpt[2]).to(dev)
        print(f"-- resuming after {start_row} of {n_base} base rows (checkpoint of the same metric / k / row counts / input files)")
    else:
        _drop_checkpoint(data_dir)

    todo = [(off, min(batch_size, n_base - off)) for off in range(start_row, n_base, batch_size)]

    def resident_batches():
        """(first row, rows) per base batch, with the batch loaded into the engine.  Parquet-backed rows are streamed piece by piece: the row
        groups of all remaining batches are decoded in order by reader threads and handed to the GPU as they arrive (cu_knn.stream_tiles),
        so decode, upload and -- across batches -- search overlap; in-memory rows are sliced and uploaded one batch ahead."""
        if isinstance(base_rows, ParquetRows):
            for i in stream_tiles(eng, base_rows, todo, follow=lambda i: (queries, k, metric, todo[i][0])):
                yield todo[i]
            return

        def batches():
            for off, n in todo:
                t = base_rows.slice(off, n)
                yield off, len(t), table_input(t), t
        with Prefetch(batches, slots=PREFETCH_SLOTS) as feed:
            for off, n, base, table in feed:
                engine_set_base(eng, base)
                del base, table
                yield off, n

    pending_write = None
    try:
        with ThreadPoolExecutor(max_workers=1) as writer:
            for batch_offset, rows in resident_batches():
                assert (k <= rows)                                                # cu_knn_ds.py:203
                if eng.search_attached:                                           # the search walked the batch while it arrived
                    d, i = eng.search_collect(out_device=True)
                else:
                    d, i = engine_search(eng, queries, k, metric, index_offset=batch_offset, out_device=True)   # + batch_offset, cu_knn_ds.py:216
                if run_d is None:
                    run_d, run_i = d, i
                else:                                                             # fold on the device: [2,Q,k] -> [Q,k]
                    run_d, run_i = eng.merge(torch.stack([run_d, d]), torch.stack([run_i, i]), k)
                del d, i
                if resume and batch_offset + rows < n_base:
                    if pending_write is not None:
                        pending_write.result()                                    # one checkpoint in flight at a time
                    tag = _checkpoint_tag(metric, k, query_count, n_base, batch_offset + rows, inputs)
                    pending_write = writer.submit(_write_checkpoint, data_dir, run_d.cpu().numpy(), run_i.cpu().numpy(), tag)
            if pending_write is not None:
                pending_write.result()
    finally:
        eng.set_prescreen(False)
        cleanup()
    assert run_d is not None and len(run_d) == query_count
    stream_cudf_to_parquet(get_partial_distances_filename(data_dir, -1), run_d.cpu().numpy(), PARQUET_ROW_GROUP)
    stream_cudf_to_parquet(get_partial_indices_filename(data_dir, -1), run_i.cpu().numpy(), PARQUET_ROW_GROUP)
    _drop_checkpoint(data_dir)
