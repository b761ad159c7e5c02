"""Seeded synthetic source datasets -- the offline stand-in for stages 1-2 of ``nw``.

The reference embeds SQuAD questions and Wikipedia passages through remote model APIs (generate_dataset.py:264-368,
model_generator.py) -- network-only, out of scope.  What the kNN stage consumes is just a parquet file with metadata
columns plus one float32 column per dimension, ``embedding_0..embedding_{d-1}`` (ParquetStreamer,
generate_dataset.py:219-262; tests/test_02_cu_knn.py:15-22).  This module writes that schema from a counter-based
generator whose values are exactly representable (24-bit integers x 2^-23), identical to the on-device generator
(nwknn_generate) and to oracle.cpu_oracle.gen_numpy.
"""
from __future__ import annotations

import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

SEED_BASE = 0x6E770001
SEED_QUERY = 0x6E770002
_MUL_A = np.uint64(0xBF58476D1CE4E5B9)
_MUL_B = np.uint64(0x94D049BB133111EB)
_ROW_STEP = np.uint64(0x9E3779B97F4A7C15)
_COL_STEP = np.uint64(0xD1342543DE82EF95)


def _finalize(z):
    z = (z ^ (z >> np.uint64(30))) * _MUL_A
    z = (z ^ (z >> np.uint64(27))) * _MUL_B
    return z ^ (z >> np.uint64(31))


def _raw24(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """24-bit unsigned draw for every (row, col): top bits of a double splitmix64 finaliser."""
    with np.errstate(over="ignore"):
        r = np.arange(row0, row0 + rows, dtype=np.uint64).reshape(-1, 1)
        c = np.arange(1, d + 1, dtype=np.uint64).reshape(1, -1)
        u = _finalize(_finalize(np.uint64(seed) + r * _ROW_STEP) + c * _COL_STEP)
    return (u >> np.uint64(40)).astype(np.int64)


def gen_rows(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """float32 [rows, d], uniform on the 2^-23 grid of [-1, 1)."""
    return ((_raw24(seed, row0, rows, d) - (1 << 23)).astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)


def gen_unit_rows(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """Non-negative rows scaled to unit L2 norm (the shape of the reference's own test data: np.random.rand +
    normalize_vector, tests/test_02_cu_knn.py:24-28), with an exact integer sum of squares so the result does not
    depend on the machine's reduction order."""
    v = _raw24(seed, row0, rows, d)
    ss = (v * v).sum(axis=1, keepdims=True)
    return (v.astype(np.float64) / np.sqrt(ss.astype(np.float64))).astype(np.float32)


def write_source_parquet(filename: str, seed: int, rows: int, d: int, normalize: bool = False, chunk_rows: int = 50000,
                         column_prefix: str = "embedding_", metadata_columns: bool = True, layout: str = "columns") -> str:
    """One parquet file, metadata columns first (unless ``metadata_columns=False``: the ``ck`` source files hold the token
    columns only, colbert_knn.py:229-233) then ``<prefix>0..d-1`` as float32, written in row groups.  ``layout="list"`` writes ONE
    ``FixedSizeList<float32, d>`` column named ``<prefix without the trailing underscore>`` instead (row-major values, so a
    tile goes to the GPU with no column gather at all); the kNN stage reads either."""
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    writer = None
    try:
        for r0 in range(0, rows, chunk_rows):
            n = min(chunk_rows, rows - r0)
            x = (gen_unit_rows if normalize else gen_rows)(seed, r0, n, d)
            cols = {
                "document_id_idx": pa.array([f"{r0 + i}_0" for i in range(n)]),
                "text": pa.array(["synthetic"] * n),
                "title": pa.array(["synthetic"] * n),
                "document_id": pa.array(np.arange(r0, r0 + n, dtype=np.int64)),
            } if metadata_columns else {}
            if layout == "list":
                cols[column_prefix.rstrip("_")] = pa.FixedSizeListArray.from_arrays(pa.array(np.ascontiguousarray(x).reshape(-1)), d)
            elif layout == "columns":
                for j in range(d):
                    cols[f"{column_prefix}{j}"] = pa.array(np.ascontiguousarray(x[:, j]))
            else:
                raise ValueError(f"layout must be 'columns' or 'list', got {layout!r}")
            table = pa.table(cols)
            if writer is None:
                writer = pq.ParquetWriter(filename, table.schema)
            writer.write_table(table)
    finally:
        if writer is not None:
            writer.close()
    return filename
