"""kNN driver, pyarrow.Table flavour -- same entry points as the reference's ``neighborhoodwatch/cu_knn.py``.

    compute_knn(data_dir, model_name, dimensions, query_filename, query_count, base_filename, base_count,
                mem_tune=False, k=100, initial_batch_size=100000, max_memory_threshold=0.1, split=True)
    process_batches(data_dir, base_table, query_table, batch_count, batch_size, k, split, engine='raft')
    tune_memory(table, batch_size, max_memory_threshold, rmm)
    stream_cudf_to_parquet(filename, df, chunk_size) / cleanup(*args)

Same argument meaning, same failure behaviour (assertions at cu_knn.py:136,177,237,281) and the same on-disk
result: one ``partial/distances<i>.parquet`` / ``partial/indices<i>.parquet`` pair per base tile, k columns named
"0".."k-1", float32 / int64, row groups of 100 000 (cu_knn.py:278-285).  What changes is everything between the
parquet files and the numbers: instead of arrow -> cuDF -> DLPack -> CuPy -> pylibraft per (base tile x query
tile) with a D2H/H2D bounce per call (cu_knn.py:205-276), a base tile is uploaded once into the HIP engine and
all queries are searched against it in one call; the engine's running top-k makes the reference's query
splitting (cu_knn.py:216-225) unnecessary -- ``split`` is accepted and does not change the result.

The reference's ``engine`` argument ('raft' | 'cuvs' | 'torch') selects the distance exactly as there:
sqeuclidean, cosine, 1 - dot (cu_knn.py:243-262).

Keyword-only additions (defaults keep the reference behaviour): ``metric`` (overrides ``engine``), ``dtype``
('fp32' bit-exact | 'fp16' MFMA path), ``device`` / ``devices`` (base tiles are dealt round-robin to the listed GPUs,
one engine and one host thread per GPU -- the reference is pinned to GPU 0, cu_knn.py:65), ``resume`` (skip base tiles
whose partial pair already exists; the reference always recomputes), ``prescreen`` (fp32 results, bit for bit,
through the fp16 MFMA pre-screen + fp32 rescoring of KnnEngine.set_prescreen).
"""
from __future__ import annotations

import gc
import math
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

from .engine import KnnEngine, default_engine, mem_info, resolve_metric
from .nw_utils import (get_full_filename, get_partial_distances_filename, get_partial_indices_filename,
                       output_dimension_validity_check)
from .parquet_stream import ParquetRows, PieceReader, Prefetch

PARQUET_ROW_GROUP = 100000          # cu_knn.py:284-285
PREFETCH_SLOTS = 2                  # base tiles alive on the host at once per GPU: the one being searched + the one being decoded


# ---- parquet out ---------------------------------------------------------------------------------------
def _as_arrow_table(df) -> pa.Table:
    if isinstance(df, pa.Table):
        return df
    if isinstance(df, np.ndarray):
        if df.ndim != 2:
            raise ValueError("expected a 2-D array")
        cols = [pa.array(np.ascontiguousarray(df[:, j])) for j in range(df.shape[1])]
        return pa.Table.from_arrays(cols, names=[str(j) for j in range(df.shape[1])])      # cu_knn.py:278-279
    return pa.Table.from_pandas(df, preserve_index=False)


def stream_cudf_to_parquet(filename, df, chunk_size, metadata: Optional[dict] = None):
    """Write a frame as ONE parquet file in ``chunk_size``-row row groups (cu_knn.py:26-51).  ``df`` may be a
    numpy [rows, k] array (columns get the names "0".."k-1"), a pandas DataFrame or a pyarrow Table.  ``metadata``
    (str -> str) is added to the file's key-value metadata; readers that do not know the keys ignore them."""
    table = _as_arrow_table(df)
    if metadata:
        merged = dict(table.schema.metadata or {})
        merged.update({str(a).encode(): str(b).encode() for a, b in metadata.items()})
        table = table.replace_schema_metadata(merged)
    writer = pq.ParquetWriter(filename, table.schema)
    try:
        for start in range(0, max(len(table), 1), chunk_size):
            writer.write_table(table.slice(start, chunk_size))
    finally:
        writer.close()


stream_to_parquet = stream_cudf_to_parquet


# ---- parquet in ----------------------------------------------------------------------------------------
def load_table(data_dir, filename, start, end):
    """cu_knn.py:112-114 (reads the whole file, then slices)."""
    return pq.read_table(get_full_filename(data_dir, filename)).slice(start, end)


def drop_columns(table, keep_columns):
    keep = set(keep_columns)
    return table.select([name for name in table.schema.names if name in keep])             # cu_knn.py:117-123


def get_embedding_count(table):
    return sum(1 for name in table.schema.names if name.startswith("embedding_"))


def list_embedding_field(schema, prefix: str = "embedding_"):
    """The single ``FixedSizeList<float, d>`` embedding column of a schema (named like the per-dimension columns without the
    index: ``embedding`` / ``token_embedding``), or None.  SURVEY N2's optional layout: the values buffer of such a column IS the
    row-major [rows, d] matrix, so nothing has to be gathered from d columns."""
    name = prefix.rstrip("_")
    if name in schema.names:
        t = schema.field(name).type
        if pa.types.is_fixed_size_list(t) and pa.types.is_floating(t.value_type):
            return schema.field(name)
    return None


def float_row_bytes(fields) -> int:
    """Bytes one row of the float columns occupies (the footprint tune_memory reasons about): scalar float columns plus
    FixedSizeList<float, d> columns."""
    total = 0
    for f in fields:
        if pa.types.is_floating(f.type):
            total += f.type.bit_width // 8
        elif pa.types.is_fixed_size_list(f.type) and pa.types.is_floating(f.type.value_type):
            total += f.type.list_size * (f.type.value_type.bit_width // 8)
    return total or 4


def table_list_matrix(table, field) -> np.ndarray:
    """Row-major float [rows, d] view of a FixedSizeList column: zero-copy for a one-chunk table, one concatenation otherwise."""
    d = field.type.list_size
    parts = []
    for ch in table.column(field.name).chunks:
        if len(ch) == 0:
            continue
        if ch.null_count:
            raise ValueError(f"column {field.name!r} holds null embeddings")
        flat = ch.flatten().to_numpy(zero_copy_only=False)        # the chunk's own rows only (offset-aware)
        parts.append(flat.reshape(len(ch), d))
    if not parts:
        return np.empty((0, d), dtype=np.float32)
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


def prep_table(data_dir, model_name, filename, count, dimension, lazy: bool = False):
    """Load, slice to ``count`` rows, check the number of embedding_* columns, keep only those (cu_knn.py:132-148).
    ``lazy=True`` (how compute_knn opens the BASE file) reads the file's metadata only and returns a ``ParquetRows`` window with the
    same ``len`` / ``schema`` / ``slice`` face: base tiles are then decoded from their own row groups, one tile ahead of the GPU
    (process_batches), instead of the whole file up front -- the reference's read is the wall clock of its kNN stage at C2 size."""
    if lazy:
        table = ParquetRows(get_full_filename(data_dir, filename), count)
    else:
        table = load_table(data_dir, filename, 0, count)
    lst = list_embedding_field(table.schema)
    total_dimensions = lst.type.list_size if lst is not None else sum("embedding_" in c for c in table.column_names)
    assert output_dimension_validity_check(model_name, dimension, total_dimensions), \
        f"Expected {dimension} values, got {total_dimensions} for model {model_name} [filename: {filename}]"
    keep = [lst.name] if lst is not None else [f"embedding_{i}" for i in range(total_dimensions)]
    if lazy:
        return table.select([name for name in table.column_names if name in set(keep)])
    return drop_columns(table, keep)


def table_to_matrix(table) -> np.ndarray:
    """Float columns of an arrow table -> dense row-major float32 [rows, d] (the job of
    ``cp.from_dlpack(df.select_dtypes(float).to_dlpack()).copy(order='C')`` + ``.astype(np.float32)``,
    cu_knn.py:208-211,243)."""
    fields = [f for f in table.schema if pa.types.is_float32(f.type) or pa.types.is_float64(f.type)]
    rows = len(table)
    out = np.empty((rows, len(fields)), dtype=np.float32)
    block = 64
    for j0 in range(0, len(fields), block):
        cols = [table.column(f.name).to_numpy() for f in fields[j0:j0 + block]]
        out[:, j0:j0 + len(cols)] = np.stack(cols, axis=1)
    return out


def table_input(table):
    """What the engine is handed for an arrow table: ``("matrix", row-major [rows, d] array)`` for the FixedSizeList layout,
    ``("columns", per-dimension views)`` (table_columns) for the reference's one-column-per-dimension layout."""
    lst = list_embedding_field(table.schema) or list_embedding_field(table.schema, "token_embedding_")
    if lst is not None:
        return "matrix", table_list_matrix(table, lst)
    return "columns", table_columns(table)


def engine_set_base(eng, kind_data):
    kind, data = kind_data
    eng.set_base(data) if kind == "matrix" else eng.set_base_columns(data)


def base_geometry(schema):
    """(dimension, 'f16' | 'f32') of the embedding columns of a schema: what ``KnnEngine.base_begin`` is told before any row arrives."""
    lst = list_embedding_field(schema) or list_embedding_field(schema, "token_embedding_")
    if lst is not None:
        return lst.type.list_size, ("f16" if pa.types.is_float16(lst.type.value_type) else "f32")
    fields = [f for f in schema if pa.types.is_floating(f.type)]
    return len(fields), ("f16" if fields and all(pa.types.is_float16(f.type) for f in fields) else "f32")


def stream_tiles(eng, rows: ParquetRows, tiles, readers=None, follow=None):
    """Load base tiles ``[(first row, rows), ...]`` of a lazy parquet window into ``eng`` one after the other, each PIECE BY PIECE: the
    row groups of ALL the listed tiles are decoded in row order by a pool of reader threads (``PieceReader``: own memory-mapped handles, a
    bounded number of pieces ahead), and every decoded piece goes straight to the GPU (``base_append*``: gather into pinned memory, H2D copy
    and layout kernel asynchronous) while the next ones are decoded.  Yields the tile's position in ``tiles`` once the tile is resident
    (``base_end`` done); while the consumer searches it, the readers are already decoding the first pieces of the next tile.
    The reference reads the whole file, then uploads tile after tile (cu_knn.py:112-114,201-211).
    ``follow(i) -> (queries, k, metric, index_offset)``: the search the consumer will run on tile i, attached to the tile before its first
    piece (``KnnEngine.search_attach``) so that it walks the tile while it arrives; when ``eng.search_attached`` is true at the yield, the
    consumer takes the result with ``eng.search_collect()`` instead of searching."""
    dims, dtype = base_geometry(rows.schema)
    counts, pieces = [], []
    for off, length in tiles:
        ps = rows.pieces(off, length)
        counts.append(len(ps))
        pieces.extend((rows, pc) for pc in ps)

    def ready(t):
        """On the reader thread, right behind the decode: the numpy views of the piece and the pointer table the library takes."""
        kind, data = table_input(t)
        return t, kind, (KnnEngine.prepare_columns(data) if kind == "columns" else data)

    # a piece is hundreds of megabytes of arrow buffers: releasing them (munmap) is work too, and not the uploading thread's
    with PieceReader(pieces, readers=readers, transform=ready) as reader, ThreadPoolExecutor(max_workers=1, thread_name_prefix="nw-drop") as dropper:
        for i, ((off, length), n_pieces) in enumerate(zip(tiles, counts)):
            eng.base_begin(length, dims, dtype)
            if follow is not None:
                queries, k, metric, index_offset = follow(i)
                eng.search_attach(queries, k, metric, index_offset=index_offset)
            for _ in range(n_pieces):
                item = next(reader)
                if item[1] == "matrix":
                    eng.base_append(item[2])
                else:
                    eng.base_append_columns(item[2])
                dropper.submit(_drop, item)
                del item
            eng.base_end()
            yield i


def _drop(item):
    """Runs on the drop thread: the last reference to a consumed piece dies here."""
    del item


def engine_search(eng, kind_data, k, metric, index_offset=0, out_device=False):
    kind, data = kind_data
    if kind == "matrix":
        return eng.search(data, k, metric, index_offset=index_offset, out_device=out_device)
    return eng.search_columns(data, k, metric, index_offset=index_offset, out_device=out_device)


def table_columns(table):
    """The float columns of an arrow table as zero-copy numpy views: a 1-D array per column, or -- when the table is chunked
    (pq.read_table yields one chunk per parquet row group) -- a list of 1-D arrays per column, which the engine's upload
    pipeline gathers from directly.  This is all the host does to a tile: nothing is concatenated or transposed; the column ->
    row-major conversion the reference performs on the GPU per tile (cu_knn.py:205-211,229-235) happens inside the engine's
    layout kernel, which reads the columns."""
    fields = [f for f in table.schema if pa.types.is_float32(f.type) or pa.types.is_float64(f.type) or pa.types.is_float16(f.type)]
    kinds = {str(f.type) for f in fields}
    cols = []
    for f in fields:
        col = table.column(f.name)
        chunks = [ch.to_numpy(zero_copy_only=False) for ch in col.chunks if len(ch) > 0]
        if len(kinds) != 1:
            chunks = [a.astype(np.float32) for a in chunks]
        cols.append(chunks[0] if len(chunks) == 1 else chunks)
    shapes = {tuple(len(a) for a in c) if isinstance(c, list) else (len(c),) for c in cols}
    if len(shapes) > 1:                                          # columns chunked differently (concatenated tables): one array each
        cols = [np.concatenate(c) if isinstance(c, list) else c for c in cols]
    return cols


# ---- memory tuning -------------------------------------------------------------------------------------
def tune_memory(table, batch_size, max_memory_threshold, rmm=None, device: int = 0):
    """Base-tile rows from the GPU's memory size -- the arithmetic of cu_knn.py:54-109 without the probing
    allocations: footprint of a tile = rows x (bytes per row of the float columns); first jump by
    ``ceil(0.2*total/used)``, then the reference's loop, whose only reachable outcomes are
    ``num_rows`` (tile >= table), ``int(0.8*tile)`` (over the threshold) or -- because ``batch_size *= 1.2`` turns it
    into a float that ``table.slice`` rejects (cu_knn.py:102-107) -- ``int(0.8*1.2*tile)``.
    ``rmm`` is accepted for signature compatibility and unused.  Total memory comes from hipMemGetInfo
    (nwknn_mem_info) instead of NVML (cu_knn.py:62-71)."""
    num_rows = len(table)
    print(f"Tuning memory settings (num rows: {num_rows}; initial batch size: {batch_size}) ...")
    _, total_gpu_memory = mem_info(device)
    print("-- total memory:", total_gpu_memory)
    row_bytes = float_row_bytes(table.schema)
    memory_used = min(batch_size, num_rows) * row_bytes
    print(f"-- memory_used {memory_used}")
    factor = math.ceil((total_gpu_memory * .2) / memory_used)
    print(f"-- factor {factor}")
    batch_size *= factor
    if num_rows < batch_size:
        print(f"-- the calculated batch size {batch_size} is bigger than total rows {num_rows}. Use total rows as the target batch size!")
        return num_rows
    memory_used = batch_size * row_bytes
    if memory_used > max_memory_threshold * total_gpu_memory:
        batch_size = int(0.8 * batch_size)
        print(f"-- found threshold {batch_size}")
        return batch_size
    grown = batch_size * 1.2
    if num_rows < grown:
        return num_rows
    batch_size = int(0.8 * grown)
    print(f"-- max batch size {batch_size}")
    return batch_size


def cleanup(*args):
    """cu_knn.py:183-190 without RMM: drop references and collect."""
    del args
    gc.collect()


# ---- the driver ----------------------------------------------------------------------------------------
def compute_knn(data_dir,
                model_name,
                dimensions,
                query_filename,
                query_count,
                base_filename,
                base_count,
                mem_tune=False,
                k=100,
                initial_batch_size=100000,
                max_memory_threshold=0.1,
                split=True,
                *,
                engine='raft',
                metric: Optional[str] = None,
                dtype: str = 'fp32',
                device: int = 0,
                devices: Optional[Sequence[int]] = None,
                resume: bool = False,
                prescreen: bool = False):
    batch_size = initial_batch_size
    print("-- prepare query source table for brute-force KNN computation.")
    query_table = prep_table(data_dir, model_name, query_filename, query_count, dimensions)
    print("-- prepare base source table for brute-force KNN computation.")
    base_table = prep_table(data_dir, model_name, base_filename, base_count, dimensions, lazy=True)

    if mem_tune:
        batch_size = tune_memory(base_table, batch_size, max_memory_threshold, None, device)

    batch_count = math.ceil(len(base_table) / batch_size)
    assert (len(base_table) % batch_size == 0) or k <= (len(base_table) % batch_size), \
        f"Cannot generate k of {k} with only {len(base_table)} rows and batch_size of {batch_size}."      # cu_knn.py:177-178

    process_batches(data_dir, base_table, query_table, batch_count, batch_size, k, split, engine, metric=metric, dtype=dtype,
                    device=device, devices=devices, resume=resume, prescreen=prescreen)


def _tile_tag(metric: str, dtype: str, batch_offset: int, batch_length: int) -> dict:
    """What a partial pair was computed with; written into both files, compared by ``resume``.  The directory name pins
    q, b and k (nw_utils.py:62-73) but not the distance, the arithmetic or the tiling."""
    return {"nwknn.metric": metric, "nwknn.dtype": "fp16" if dtype in ("fp16", "f16") else "fp32",
            "nwknn.base_rows": f"{batch_offset}+{batch_length}"}


def _partial_pair_exists(data_dir, i, rows, k=None, tag: Optional[dict] = None) -> bool:
    """A finished pair for base tile ``i``: both files readable, ``rows`` rows and (when given) ``k`` columns each, and --
    when ``tag`` is given -- written by a run with the same metric, dtype and tile bounds.  Files without the tag (written
    by the reference or by an older build) do not match a tag, so they are recomputed rather than trusted."""
    fi, fd = get_partial_indices_filename(data_dir, i), get_partial_distances_filename(data_dir, i)
    try:
        if not (os.path.getsize(fi) > 0 and os.path.getsize(fd) > 0):
            return False
        for f in (fi, fd):
            pf = pq.ParquetFile(f)
            if pf.metadata.num_rows != rows or (k is not None and pf.metadata.num_columns != k):
                return False
            if tag is not None:
                kv = pf.schema_arrow.metadata or {}
                if any(kv.get(a.encode()) != b.encode() for a, b in tag.items()):
                    return False
        return True
    except (OSError, pa.ArrowInvalid):
        return False


def process_batches(data_dir,
                    base_table,
                    query_table,
                    batch_count,
                    batch_size,
                    k,
                    split,
                    engine='raft',
                    *,
                    metric: Optional[str] = None,
                    dtype: str = 'fp32',
                    device: int = 0,
                    devices: Optional[Sequence[int]] = None,
                    resume: bool = False,
                    prescreen: bool = False):
    """One partial (distances, indices) parquet pair per base tile (cu_knn.py:193-287)."""
    metric = resolve_metric(metric, engine)
    devices = list(devices) if devices else [device]
    queries = table_input(query_table)                          # zero-copy views of the arrow columns (or of the one list column)

    def run_tiles(slot: int):
        # tiles slot, slot+n, slot+2n, ... on GPU devices[slot]; engines are per (device, slot) so that two slots may share a device
        eng: KnnEngine = default_engine(devices[slot]) if len(devices) == 1 else KnnEngine(devices[slot])
        eng.set_compute_dtype(dtype)
        eng.set_prescreen(prescreen)

        def plan():
            """This slot's tiles: (tile ordinal, first row, rows, provenance tag) -- minus the ones a resumed run already holds."""
            out = []
            for start in range(slot, batch_count, len(devices)):
                batch_offset = start * batch_size
                batch_length = batch_size if start != batch_count - 1 else len(base_table) - batch_offset
                tag = _tile_tag(metric, dtype, batch_offset, batch_length)
                if resume and _partial_pair_exists(data_dir, start, len(query_table), k, tag):
                    print(f"-- base tile {start}: partial files of the same metric/dtype/tile present, skipped")
                    continue
                out.append((start, batch_offset, batch_length, tag))
            return out

        def resident_tiles():
            """(tile ordinal, first row, rows, tag, dimension) with the tile loaded into the engine.  A lazy parquet window (how compute_knn
            opens the base) is streamed piece by piece (stream_tiles); a pyarrow.Table -- process_batches called as the reference calls it
            -- is sliced for free and uploaded tile by tile, the next slice prepared one tile ahead."""
            todo = plan()
            if isinstance(base_table, ParquetRows):
                dims = base_geometry(base_table.schema)[0]
                for i in stream_tiles(eng, base_table, [(off, length) for _, off, length, _ in todo],
                                      follow=lambda i: (queries, k, metric, todo[i][1])):
                    start, batch_offset, batch_length, tag = todo[i]
                    yield start, batch_offset, batch_length, tag, dims
                return

            def tiles():
                for start, batch_offset, batch_length, tag in todo:
                    tile = base_table.slice(batch_offset, batch_length)
                    yield start, batch_offset, batch_length, tag, table_input(tile), tile
            with Prefetch(tiles, slots=PREFETCH_SLOTS) as feed:
                for start, batch_offset, batch_length, tag, dataset, tile in feed:
                    dims = dataset[1].shape[1] if dataset[0] == "matrix" else len(dataset[1])
                    engine_set_base(eng, dataset)
                    del dataset, tile
                    yield start, batch_offset, batch_length, tag, dims

        def write_pair(start, distances, indices, tag):
            # the two files of a pair side by side (pyarrow releases the GIL while it encodes): the last tile's pair is the tail of the stage
            return [writer.submit(stream_cudf_to_parquet, get_partial_distances_filename(data_dir, start), distances, PARQUET_ROW_GROUP, tag),
                    writer.submit(stream_cudf_to_parquet, get_partial_indices_filename(data_dir, start), indices, PARQUET_ROW_GROUP, tag)]

        writes = []
        try:
            with ThreadPoolExecutor(max_workers=2) as writer:      # the pair of tile i is written while tile i+1 is loaded and searched
                for start, batch_offset, batch_length, tag, dims in resident_tiles():
                    assert (k <= batch_length)                              # cu_knn.py:237
                    print(f"Dataset shape: ({batch_length}, {dims}), Query shape: ({len(query_table)}, {dims})")
                    if eng.search_attached:                                 # the search walked the tile while it arrived
                        distances, indices = eng.search_collect()
                    else:
                        distances, indices = engine_search(eng, queries, k, metric, index_offset=batch_offset)   # + batch_offset, cu_knn.py:273
                    assert (len(distances) == len(query_table))             # cu_knn.py:281-282
                    assert (len(indices) == len(query_table))
                    writes.extend(write_pair(start, distances, indices, tag))
                    del distances, indices
                for w in writes:
                    w.result()
        finally:
            eng.set_compute_dtype('fp32')
            eng.set_prescreen(False)
            if len(devices) > 1:
                eng.close()
            gc.collect()

    if len(devices) == 1:
        run_tiles(0)
    else:
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:      # ctypes releases the GIL during the GPU calls
            for f in [pool.submit(run_tiles, s) for s in range(len(devices))]:
                f.result()
