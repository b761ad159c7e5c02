"""kNN driver, pyarrow.dataset flavour -- same entry point as the reference's ``neighborhoodwatch/cu_knn_ds.py``.

    compute_knn_ds(data_dir, dimensions, query_filename, query_count, base_filename, base_count,
                   mem_tune=False, k=100, initial_batch_size=200000, max_memory_threshold=0.2, split=True)

The base file is streamed in record batches of ``batch_size`` rows (cu_knn_ds.py:190); each batch is searched on
the GPU with global row ids and folded into the running result with the device merge kernel, and
``partial/final_distances.parquet`` / ``final_indices.parquet`` are written ONCE at the end.  (The reference writes
the final files inside the loop, cu_knn_ds.py:232-233, so every base batch overwrites the previous one and the
result is only right for a single batch; merge then finds no partial files and is a no-op, merge.py:40-42.)
"""
from __future__ import annotations

import math

import numpy as np
import pyarrow as pa
import pyarrow.dataset as ds

from .cu_knn import PARQUET_ROW_GROUP, cleanup, stream_cudf_to_parquet, table_columns
from .engine import default_engine, mem_info, resolve_metric
from .nw_utils import get_full_filename, get_partial_distances_filename, get_partial_indices_filename


def get_gpu_memory(device: int = 0):
    """Total HBM bytes of one GPU (cu_knn_ds.py:30-36 asks NVML for card 0)."""
    return mem_info(device)[1]


def load_dataset(data_dir, filename):
    return ds.dataset(get_full_filename(data_dir, filename), format="parquet")


def dataset_count(dataset):
    return dataset.scanner(batch_size=100000).count_rows()


def slice_dataset(dataset, start, end):
    return ds.dataset(dataset.take(np.arange(start, min(end, dataset_count(dataset)))))


def get_dataset_columns(dataset, n):
    """Columns handed to the scanner: the text column if present plus embedding_0..n-1 (cu_knn_ds.py:122-133);
    only the float columns reach the GPU."""
    names = dataset.schema.names
    cols = [c for c in ("text", "question") if c in names][-1:]
    return cols + [f"embedding_{i}" for i in range(n)]


def tune_memory(dataset, batch_size, max_memory_threshold, rmm, column_names, device: int = 0):
    """cu_knn_ds.py:57-109 as arithmetic (see cu_knn.tune_memory): here the first jump already uses
    ``max_memory_threshold`` (cu_knn_ds.py:69), so the loop's first probe sits at the threshold."""
    num_rows = dataset_count(dataset)
    total = get_gpu_memory(device)
    print("-- total memory per GPU:", total)
    fields = [dataset.schema.field(c) for c in column_names if c in dataset.schema.names]
    row_bytes = sum(f.type.bit_width // 8 for f in fields if pa.types.is_floating(f.type)) or 4
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
                   prescreen: bool = False):
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
                            query_count, k, split, engine=engine, metric=metric, device=device, prescreen=prescreen)


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
                            prescreen: bool = False):
    metric = resolve_metric(metric, engine)
    eng = default_engine(device)
    eng.set_prescreen(prescreen)
    query_table = query_dataset.to_table(columns=query_column_names)
    queries = table_columns(query_table)
    assert len(query_table) == query_count                                        # cu_knn_ds.py:229-230
    run_d = run_i = None
    batch_offset = 0
    pending, pending_rows = [], 0

    def flush():
        nonlocal run_d, run_i, batch_offset, pending, pending_rows
        base_table = pa.Table.from_batches(pending)
        base = table_columns(base_table)
        assert (k <= len(base_table))                                             # cu_knn_ds.py:203
        eng.set_base_columns(base)
        d, i = eng.search_columns(queries, k, metric, index_offset=batch_offset)  # + batch_offset, cu_knn_ds.py:216
        if run_d is None:
            run_d, run_i = d, i
        else:
            run_d, run_i = eng.merge(np.stack([run_d, d]), np.stack([run_i, i]), k)
        batch_offset += len(base_table)
        pending, pending_rows = [], 0
        cleanup(base)

    # the scanner may cut record batches at row-group boundaries; regroup to exactly batch_size rows
    for rb in base_dataset.to_batches(batch_size=batch_size, columns=base_column_names):
        while rb.num_rows:
            take = min(batch_size - pending_rows, rb.num_rows)
            pending.append(rb.slice(0, take)); pending_rows += take
            rb = rb.slice(take)
            if pending_rows == batch_size:
                flush()
    if pending_rows:
        flush()
    eng.set_prescreen(False)
    assert run_d is not None and len(run_d) == query_count
    stream_cudf_to_parquet(get_partial_distances_filename(data_dir, -1), run_d, PARQUET_ROW_GROUP)
    stream_cudf_to_parquet(get_partial_indices_filename(data_dir, -1), run_i, PARQUET_ROW_GROUP)
