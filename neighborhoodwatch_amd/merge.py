"""Merge of the per-base-tile partial tables -- same entry point as the reference's ``neighborhoodwatch/merge.py``.

    merge_indices_and_distances(data_dir, k=100)

Reads ``<data_dir>/partial/indices<i>.parquet`` / ``distances<i>.parquet`` for i = 0..n-1 (discovered exactly
like merge.py:15-27: files matching ``indices(\\d+)\\.parquet``), keeps for every query row the k smallest
distances over all partial tables, ascending, and writes ``partial/final_indices.parquet`` /
``final_distances.parquet`` with the first partial file's column names ("0".."k-1").

The reference folds the tables pairwise on the CPU with ``np.argsort`` over [Q,2k] blocks (merge.py:73-101).  The
fold of sorted lists keeping k equals the k smallest of the concatenation, so here all tables of a query block go
to the GPU at once (nwknn_merge: LDS bitonic sort by (distance, index)).  Differences, all deliberate:
  * equal distances: the reference's order is whatever numpy's unstable introsort yields; here lower index wins;
  * Q > 10 000 000: merge.py:61-65 slices with ``start`` instead of ``start*batch_size`` and mis-merges; here every
    batch is sliced at its true offset;
  * ``RowNum`` columns (tests/test_03_merge.py:14-24): merge.py:33-34 calls ``remove_column`` and discards the result,
    so the column would be merged as data; here it is dropped.
"""
from __future__ import annotations

import glob
import math
import re

import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq

from .engine import default_engine
from .nw_utils import get_partial_distances_filename, get_partial_indices_filename

MERGE_QUERY_BATCH = 10_000_000       # merge.py:46


def get_file_count(data_dir):
    """Number of files ``<data_dir>/indices<digits>.parquet`` (merge.py:15-27; called with ``<data_dir>/partial``)."""
    pattern = re.compile(re.escape(str(data_dir)) + r"/indices(\d+)\.parquet")
    return sum(1 for name in glob.glob(f"{data_dir}/indices*.parquet") if pattern.match(name))


def read_ifvec_parquet_with_proper_schema(filename):
    table = pq.read_table(filename)
    at = table.schema.get_field_index("RowNum")
    return table.remove_column(at) if at != -1 else table


def _matrix(table, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.stack([table.column(j).to_numpy() for j in range(table.num_columns)], axis=1), dtype=dtype)


def merge_partial_arrays(dists, idx, k: int, device: int = 0):
    """dists f32 [n,Q,k_in], idx i64 [n,Q,k_in] -> ([Q,k], [Q,k]) on the GPU (KnnEngine.merge folds in groups when
    n*k_in exceeds what one LDS sort holds)."""
    n, q, kin = np.shape(dists)
    return default_engine(device).merge(dists, idx, min(k, n * kin))


def merge_indices_and_distances(data_dir, k=100, *, device: int = 0):
    file_count = get_file_count(f"{data_dir}/partial")
    if file_count <= 0:
        return                                                        # merge.py:40-42: nothing to do
    first_i = read_ifvec_parquet_with_proper_schema(get_partial_indices_filename(data_dir, 0))
    first_d = read_ifvec_parquet_with_proper_schema(get_partial_distances_filename(data_dir, 0))
    n_rows = len(first_i)
    batch_size = min(MERGE_QUERY_BATCH, n_rows)
    batch_count = math.ceil(n_rows / batch_size) if batch_size else 0
    idx_tables = [first_i] + [read_ifvec_parquet_with_proper_schema(get_partial_indices_filename(data_dir, j)) for j in range(1, file_count)]
    dst_tables = [first_d] + [read_ifvec_parquet_with_proper_schema(get_partial_distances_filename(data_dir, j)) for j in range(1, file_count)]
    idx_names = first_i.schema.names[:k]
    dst_names = first_d.schema.names[:k]
    iw = dw = None
    try:
        for b in range(batch_count):
            lo = b * batch_size
            di = np.stack([_matrix(t.slice(lo, batch_size), np.int64) for t in idx_tables])
            dd = np.stack([_matrix(t.slice(lo, batch_size), np.float32) for t in dst_tables])
            if file_count == 1:
                fd, fi = dd[0][:, :k], di[0][:, :k]                    # merge.py:67-69: a single table is passed through
            else:
                fd, fi = merge_partial_arrays(dd, di, k, device)
                assert (np.diff(fd, axis=1) >= 0).all()                # merge.py:104: rows ascending
            ti = pa.Table.from_pandas(pd.DataFrame(fi, columns=idx_names[:fi.shape[1]]))      # merge.py:110-111
            tdd = pa.Table.from_pandas(pd.DataFrame(fd, columns=dst_names[:fd.shape[1]]))
            if iw is None:
                iw = pq.ParquetWriter(get_partial_indices_filename(data_dir, -1), ti.schema)
                dw = pq.ParquetWriter(get_partial_distances_filename(data_dir, -1), tdd.schema)
            iw.write_table(ti)
            dw.write_table(tdd)
    finally:
        if iw is not None:
            iw.close()
        if dw is not None:
            dw.close()
