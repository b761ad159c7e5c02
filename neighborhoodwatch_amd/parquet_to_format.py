"""Stage 5 of ``nw``: parquet -> ``.fvec`` / ``.ivec`` (+ optional hdf5) and the ``1 - dot == distance / 2`` validator.
Same entry points as the reference's ``neighborhoodwatch/parquet_to_format.py``.

File format (parquet_to_format.py:71-89, read back by :37-69 and misc/ifvec_reader.py): for every vector a little
endian int32 ``dim`` followed by ``dim`` 4-byte values -- float32 for ``.fvec``, int32 for ``.ivec`` (neighbour ids
are narrowed from int64 to int32, parquet_to_format.py:84).

The reference writes row by row through ``DataFrame.iterrows`` (197 s for 4 M rows in docs/cli.png, twice the kNN
stage); here a whole block of rows is laid out as one ``[rows, 1 + dim]`` int32 image and written with a single
``tofile``.  The byte stream is identical (tests/golden/ifvec_*.npz were produced by the reference's own writer).
hdf5 output needs ``h5py``; when it is not installed the hdf5 step is skipped with a message, everything else runs.
"""
from __future__ import annotations

import os
import struct

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

from .nw_utils import get_full_filename, output_dimension_validity_check

WRITE_BLOCK_ROWS = 262144


# ---- names (nw_utils.py:110-154 in the reference) -------------------------------------------------------
def get_ivec_fvec_filenames(knn_model_data_homedir, model_name, dimensions, base_count, query_count, k, output_dtype=None):
    stem = f"{model_name.replace('/', '_')}_{dimensions}" + (f"_{output_dtype}" if output_dtype is not None else "")
    names = (f"{stem}_query_vectors_{query_count}.fvec", f"{stem}_base_vectors_{base_count}.fvec",
             f"{stem}_indices_b{base_count}_q{query_count}_k{k}.ivec", f"{stem}_distances_b{base_count}_q{query_count}_k{k}.fvec")
    return tuple(get_full_filename(knn_model_data_homedir, n) for n in names)


def get_hdf5_filename(knn_model_data_homedir, model_name, dimensions, base_count, query_count, k, output_dtype=None):
    stem = f"{model_name.replace('/', '_')}_{dimensions}" + (f"_{output_dtype}" if output_dtype is not None else "")
    return get_full_filename(knn_model_data_homedir, f"{stem}_base_{base_count}_query_{query_count}_k{k}.hdf5")


# ---- readers ---------------------------------------------------------------------------------------------
def count_vectors(data_dir, filename):
    """Number of vectors in an fvec/ivec file (parquet_to_format.py:37-48 walks the records; every record of a file
    written here has the same dim, so the count follows from the size -- ragged files fall back to the walk)."""
    full = get_full_filename(data_dir, filename)
    size = os.path.getsize(full)
    if size == 0:
        return 0
    with open(full, "rb") as f:
        dim = struct.unpack("<i", f.read(4))[0]
        rec = 4 * (1 + dim)
        if size % rec == 0:
            f.seek(size - rec)
            if struct.unpack("<i", f.read(4))[0] == dim:
                return size // rec
        f.seek(0)
        count = 0
        while True:
            head = f.read(4)
            if not head:
                return count
            f.seek(4 * struct.unpack("<i", head)[0], 1)
            count += 1


def get_nth_vector(data_dir, filename, n):
    """Tuple of the n-th vector (parquet_to_format.py:55-68); floats for .fvec, ints for .ivec."""
    full = get_full_filename(data_dir, filename)
    dtype = np.int32 if full.endswith("ivec") else np.float32
    with open(full, "rb") as f:
        dim = struct.unpack("<i", f.read(4))[0]
        f.seek(int(4 * n * (1 + dim)), 1)
        assert os.path.getsize(full) >= f.tell() + 4 * dim, "file size is less than expected"
        return tuple(np.frombuffer(f.read(4 * dim), dtype=dtype).tolist())


def get_first_vector(data_dir, filename):
    return get_nth_vector(data_dir, filename, 0)


def read_vectors(data_dir, filename) -> np.ndarray:
    """Whole fvec/ivec file as an array [count, dim] (the vectorised form of misc/ifvec_reader.py)."""
    full = get_full_filename(data_dir, filename)
    raw = np.fromfile(full, dtype=np.int32)
    if raw.size == 0:
        return raw.reshape(0, 0)
    dim = int(raw[0])
    body = raw.reshape(-1, 1 + dim)[:, 1:]
    return body if full.endswith("ivec") else body.view(np.float32)


# ---- writer ----------------------------------------------------------------------------------------------
def write_ivec_fvec_from_dataframe(data_dir, model_name, filename, df, type_char, num_columns):
    """``df``: pandas DataFrame, pyarrow Table or 2-D array; one record per row (parquet_to_format.py:71-89).  A trailing
    extra column (RowNum) is dropped exactly as there."""
    if hasattr(df, "to_numpy"):
        values = df.to_numpy()
    elif hasattr(df, "to_pandas"):
        values = df.to_pandas().to_numpy()
    else:
        values = np.asarray(df)
    if values.ndim != 2:
        raise ValueError("expected a 2-D table")
    if values.shape[1] == num_columns + 1:
        values = values[:, :-1]
    assert output_dimension_validity_check(model_name, num_columns, values.shape[1]), \
        f"Expected {num_columns} values, got {values.shape[1]} for model {model_name} [filename: {filename}]"
    dim = values.shape[1]
    full = get_full_filename(data_dir, filename)
    with open(full, "wb") as f:
        for r0 in range(0, values.shape[0], WRITE_BLOCK_ROWS):
            block = values[r0:r0 + WRITE_BLOCK_ROWS]
            image = np.empty((block.shape[0], 1 + dim), dtype="<i4")
            image[:, 0] = dim
            if type_char == "f":
                image[:, 1:] = np.ascontiguousarray(block, dtype="<f4").view("<i4")
            else:
                image[:, 1:] = block.astype("<i4")
            image.tofile(f)


def read_parquet_to_dataframe(data_dir, filename):
    return pq.read_table(get_full_filename(data_dir, filename)).to_pandas()


def read_and_extract(data_dir, input_parquet, rowcount, dimensions, column_names=None):
    table = pq.read_table(get_full_filename(data_dir, input_parquet)).slice(0, rowcount)
    if column_names is None and "embedding" in table.schema.names and pa.types.is_fixed_size_list(table.schema.field("embedding").type):
        from .cu_knn import table_list_matrix                                     # the one-column FixedSizeList layout
        return table_list_matrix(table, table.schema.field("embedding"))
    if column_names is None:
        keep = {f"embedding_{i}" for i in range(dimensions)}
        table = table.select([n for n in table.schema.names if n in keep])      # parquet_to_format.py:97-105
    return table.to_pandas()


def is_empty_file(filename):
    return not os.path.exists(filename) or os.path.getsize(filename) == 0


def write_hdf5(data_dir, model_name, df, filename, group, output_dtype=None):
    """Append ``df`` as dataset ``group`` of an hdf5 file (parquet_to_format.py:322-348); needs h5py."""
    try:
        import h5py
    except ImportError:
        print(f"h5py is not installed: skipping hdf5 group '{group}' of {filename}")
        return False
    with h5py.File(get_full_filename(data_dir, filename), "a") as f:
        if group in f:
            print(f"Group '{group}' already exists in file '{filename}'")
        elif output_dtype is None:
            f.create_dataset(group, data=df.values)
        else:
            assert model_name.startswith("voyage")
            t = {"float": np.float32, "int8": np.int8, "binary": np.int8, "uint8": np.uint8, "ubinary": np.uint8}[output_dtype]
            ds = f.create_dataset(group, data=df.values, dtype=t)
            if output_dtype in ("binary", "ubinary"):
                ds.attrs["encoding"] = "binary_int8" if output_dtype == "binary" else "binary_uint8"
    return True


def _emit(data_dir, model_name, target, df, type_char, width, output_hdf5, hdf5_file, group):
    if is_empty_file(target):
        write_ivec_fvec_from_dataframe(data_dir, model_name, target, df, type_char, width)
    else:
        print(f"File {target} already exists")                                   # parquet_to_format.py:130-135
    if output_hdf5:
        write_hdf5(data_dir, model_name, df, hdf5_file, group)


def generate_query_vectors_fvec(data_dir, model_name, input_parquet, query_count, dimensions, query_vectors_fvec_file, output_hdf5=True,
                                column_names=None, hdf5_file=None):
    _emit(data_dir, model_name, query_vectors_fvec_file, read_and_extract(data_dir, input_parquet, query_count, dimensions, column_names),
          "f", dimensions, output_hdf5, hdf5_file, "test")


def generate_base_vectors_fvec(data_dir, model_name, input_parquet, base_count, dimensions, base_vectors_fvec_file, output_hdf5=True,
                               column_names=None, hdf5_file=None):
    _emit(data_dir, model_name, base_vectors_fvec_file, read_and_extract(data_dir, input_parquet, base_count, dimensions, column_names),
          "f", dimensions, output_hdf5, hdf5_file, "train")


def generate_distances_fvec(data_dir, model_name, input_parquet, k, distances_fvec_file, output_hdf5=True, hdf5_file=None):
    _emit(data_dir, model_name, distances_fvec_file, read_parquet_to_dataframe(data_dir, input_parquet), "f", k, output_hdf5, hdf5_file, "distances")


def generate_indices_ivec(data_dir, model_name, input_parquet, k, indices_fvec_file, output_hdf5=True, hdf5_file=None):
    _emit(data_dir, model_name, indices_fvec_file, read_parquet_to_dataframe(data_dir, input_parquet), "i", k, output_hdf5, hdf5_file, "neighbors")


def generate_output_files(data_dir, model_name, dimensions, base_vectors_parquet, query_vectors_parquet, base_count, query_count,
                          final_indices_parquet, final_distances_parquet, k, output_hdf5=True, column_names=None, output_dtype=None):
    """parquet_to_format.py:213-319: writes the four files (and the hdf5 groups test/train/neighbors/distances) and
    returns ``(query_vector_fvec, base_vector_fvec, indices_ivec, distances_fvec)``."""
    qf, bf, iif, df_ = get_ivec_fvec_filenames(data_dir, model_name, dimensions, base_count, query_count, k, output_dtype)
    h5 = get_hdf5_filename(data_dir, model_name, dimensions, base_count, query_count, k, output_dtype)
    generate_query_vectors_fvec(data_dir, model_name, query_vectors_parquet, query_count, dimensions, qf, output_hdf5, column_names, h5)
    print(f"{qf} - query vector count: {count_vectors(data_dir, qf)}, dimensions: {len(get_first_vector(data_dir, qf))}")
    generate_base_vectors_fvec(data_dir, model_name, base_vectors_parquet, base_count, dimensions, bf, output_hdf5, column_names, h5)
    print(f"{bf} - base vector count: {count_vectors(data_dir, bf)}, dimensions: {len(get_first_vector(data_dir, bf))}")
    generate_indices_ivec(data_dir, model_name, final_indices_parquet, k, iif, output_hdf5, h5)
    print(f"{iif} - indices count: {count_vectors(data_dir, iif)}, k: {len(get_first_vector(data_dir, iif))}")
    generate_distances_fvec(data_dir, model_name, final_distances_parquet, k, df_, output_hdf5, h5)
    print(f"{df_} - distances count: {count_vectors(data_dir, df_)}, k: {len(get_first_vector(data_dir, df_))}")
    return qf, bf, iif, df_


# ---- validator -------------------------------------------------------------------------------------------
def dot_product(A, B):
    return sum(a * b for a, b in zip(A, B))


def _report(n, col, x, dist, idx):
    print(f"Expected '1 - similarity' ({x}) equal to distance ({dist[n, col]}) for query vector {n} "
          f"and base vector {idx[n, col]}")


def _validate_rows_on_host(queries, base, idx, dist, rows, cols=None) -> int:
    """The reference's check in float64 for the given query rows (all neighbours, or only ``cols[n]``): the verdict of record."""
    bad_total = 0
    for n in rows:
        q = queries[n].astype(np.float64)
        js = np.arange(idx.shape[1]) if cols is None else cols[n]
        similarity = base[idx[n, js]].astype(np.float64) @ q
        bad = ~np.isclose(1.0 - similarity, dist[n, js].astype(np.float64) / 2.0)
        for t in np.flatnonzero(bad):
            _report(n, js[t], 1 - similarity[t], dist, idx)
        bad_total += int(bad.sum())
    return bad_total


def validate_files_v0(data_dir, query_vector_fvec, base_vector_fvec, indices_ivec, distances_fvec, device=None):
    """For every query and every listed neighbour: ``isclose(1 - dot(q, b), distance / 2)`` -- squared L2 of unit vectors
    (parquet_to_format.py:351-383); returns the mismatch count (the reference only prints it).

    With an MI355X present (``device`` = its ordinal, default: GPU 0 when there is one) the gather + dot of the Q x k pairs runs on the GPU
    in float64 (nwknn_pair_dots; the reference walks the pairs row by row in Python).  A float64 sum on the GPU and numpy's differ by
    ~1e-13 (summation order), so a pair whose ``|1 - dot - distance / 2|`` is within 1e-10 of numpy.isclose's tolerance edge is handed to
    the host check of record; every other pair is decided from the GPU value.  Without a GPU the whole check runs on the host."""
    queries = read_vectors(data_dir, query_vector_fvec)
    base = read_vectors(data_dir, base_vector_fvec)
    idx = read_vectors(data_dir, indices_ivec)
    dist = read_vectors(data_dir, distances_fvec)
    nonzero = [n for n in range(queries.shape[0]) if queries[n].any()]
    for n in range(queries.shape[0]):
        if not queries[n].any():
            print(f"Skipping zero query vector {n}/{queries.shape[0]}")
    if device is None:
        from .engine import device_count
        try:
            device = 0 if device_count() > 0 else -1
        except Exception:
            device = -1
    if device < 0 or idx.shape[1] > 4096 or not nonzero:
        total_mismatch_count = _validate_rows_on_host(queries, base, idx, dist, nonzero)
        print(f"Total mismatch count: {total_mismatch_count}")
        return total_mismatch_count
    from .engine import KnnEngine
    from ._lib import NwknnError
    # The band of "too close to call" scales with the operands: a float64 sum on the GPU and numpy's differ by ~d * eps64 * sum |q_j b_j|
    # <= d * eps64 * |q| |b|, which is ~1e-13 for unit rows and grows with the norms (the reference's files hold unit rows; an unnormalised
    # set must not get a GPU verdict that the host check of record would not give).
    qn = np.sqrt((queries.astype(np.float64) ** 2).sum(axis=1))
    bmax = float(np.sqrt((base.astype(np.float64) ** 2).sum(axis=1).max())) if base.shape[0] else 0.0
    total_mismatch_count = 0
    try:
        with KnnEngine(int(device)) as eng:
            eng.set_base(np.ascontiguousarray(base, dtype=np.float32))
            step = max(1, (1 << 24) // max(idx.shape[1], 1))                         # ~16M pairs per call
            rows = np.asarray(nonzero, dtype=np.int64)
            for r0 in range(0, len(rows), step):
                rs = rows[r0:r0 + step]
                x = 1.0 - eng.pair_dots(np.ascontiguousarray(queries[rs], dtype=np.float32), idx[rs])
                half = dist[rs].astype(np.float64) / 2.0
                gap = np.abs(x - half) - (1e-8 + 1e-5 * np.abs(half))               # numpy.isclose defaults: |a - b| <= atol + rtol * |b|
                band = 1e-10 * np.maximum(1.0, qn[rs] * bmax)[:, None]
                unsure = ~(np.abs(gap) > band)                                       # too close to the edge to call (NaN included)
                bad = (gap > band) & ~unsure
                for t, col in zip(*np.nonzero(bad)):
                    _report(int(rs[t]), int(col), x[t, col], dist, idx)
                total_mismatch_count += int(bad.sum())
                for t in np.flatnonzero(unsure.any(axis=1)):
                    n = int(rs[t])
                    total_mismatch_count += _validate_rows_on_host(queries, base, idx, dist, [n], {n: np.flatnonzero(unsure[t])})
    except (NwknnError, MemoryError) as e:
        # a base larger than the GPU's free memory, a GPU busy with another job, ...: the reference's validator would have completed, so
        # the host check of record takes the whole job (and says why)
        print(f"validate_files_v0: the GPU check could not run ({e}); checking on the host")
        total_mismatch_count = _validate_rows_on_host(queries, base, idx, dist, nonzero)
    print(f"Total mismatch count: {total_mismatch_count}")
    return total_mismatch_count
