"""CPU: host-side mirror of the reference interface -- naming contract, memory-tuning arithmetic, parquet schema,
CLI surface, sharding arithmetic.  No GPU compute is invoked."""
import inspect

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from neighborhoodwatch_amd import cu_knn, cu_knn_ds, merge, nw_utils, synth
from neighborhoodwatch_amd.distributed import shard_bounds
from neighborhoodwatch_amd.engine import resolve_metric


def test_partial_and_final_filenames():
    # nw_utils.py:94-107 -- the names are the API between the kNN stage and the merge stage
    assert nw_utils.get_partial_indices_filename("/d/m/q1_b2_k3", 0) == "/d/m/q1_b2_k3/partial/indices0.parquet"
    assert nw_utils.get_partial_distances_filename("/d/m/q1_b2_k3", 12) == "/d/m/q1_b2_k3/partial/distances12.parquet"
    assert nw_utils.get_partial_indices_filename("x", -1) == "x/partial/final_indices.parquet"
    assert nw_utils.get_partial_distances_filename("x", -1) == "x/partial/final_distances.parquet"


def test_output_folder_and_source_names(tmp_path):
    d = nw_utils.setup_model_output_folder(str(tmp_path), "intfloat/e5-small-v2", 1000, 10000, 100)
    assert d == f"{tmp_path}/intfloat_e5-small-v2/q1000_b10000_k100"                         # nw_utils.py:62-73
    assert (tmp_path / "intfloat_e5-small-v2" / "q1000_b10000_k100" / "partial").is_dir()
    assert nw_utils.get_source_query_dataset_filename(d, "intfloat/e5-small-v2", 1000, 384).endswith(
        "/intfloat_e5-small-v2_384_query_vector_data_1000.parquet")                         # nw_utils.py:76-82
    assert nw_utils.get_source_base_dataset_filename(d, "voyage-3-large", 5, 1024, "int8").endswith(
        "/voyage-3-large_1024_int8_base_vector_data_5.parquet")
    assert nw_utils.get_full_filename("a/b", "c.parquet") == "a/b/c.parquet"
    assert nw_utils.get_full_filename("a/b", "a/b/c.parquet") == "a/b/c.parquet"


def test_model_dimension_table():
    # model_generator.py:61-113
    assert nw_utils.get_default_model_dimension_size("intfloat/e5-small-v2") == 384
    assert nw_utils.get_default_model_dimension_size("intfloat/e5-large-v2") == 1024
    assert nw_utils.get_default_model_dimension_size("text-embedding-ada-002") == 1536
    assert nw_utils.get_default_model_dimension_size("colbertv2.0") == 128
    assert nw_utils.get_effective_embedding_size("text-embedding-3-large", 256) == 256
    assert nw_utils.get_effective_embedding_size("intfloat/e5-base-v2", 256) == 768        # size ignored
    with pytest.raises(AssertionError):
        nw_utils.get_effective_embedding_size("text-embedding-3-small", 4096)
    assert len(nw_utils.get_valid_model_name_list()) == 15
    assert nw_utils.output_dimension_validity_check("voyage-3-large", 1024, 128)
    assert not nw_utils.output_dimension_validity_check("intfloat/e5-base-v2", 768, 96)


def test_signatures_match_the_reference():
    # cu_knn.py:151-162, :193-200, :54, cu_knn_ds.py:136-146, merge.py:39
    p = list(inspect.signature(cu_knn.compute_knn).parameters.values())
    assert [x.name for x in p[:12]] == ["data_dir", "model_name", "dimensions", "query_filename", "query_count", "base_filename",
                                        "base_count", "mem_tune", "k", "initial_batch_size", "max_memory_threshold", "split"]
    assert [x.default for x in p[7:12]] == [False, 100, 100000, 0.1, True]
    assert all(x.kind is inspect.Parameter.KEYWORD_ONLY for x in p[12:])
    p = list(inspect.signature(cu_knn.process_batches).parameters.values())
    assert [x.name for x in p[:8]] == ["data_dir", "base_table", "query_table", "batch_count", "batch_size", "k", "split", "engine"]
    assert p[7].default == "raft"
    p = list(inspect.signature(cu_knn_ds.compute_knn_ds).parameters.values())
    assert [x.name for x in p[:11]] == ["data_dir", "dimensions", "query_filename", "query_count", "base_filename", "base_count",
                                        "mem_tune", "k", "initial_batch_size", "max_memory_threshold", "split"]
    assert [x.default for x in p[6:11]] == [False, 100, 200000, 0.2, True]
    p = list(inspect.signature(merge.merge_indices_and_distances).parameters.values())
    assert [x.name for x in p[:2]] == ["data_dir", "k"] and p[1].default == 100
    assert [x.name for x in inspect.signature(cu_knn.tune_memory).parameters.values()][:4] == ["table", "batch_size", "max_memory_threshold", "rmm"]


def test_engine_names_select_the_reference_distances():
    assert resolve_metric(None, "raft") == "sqeuclidean"       # cu_knn.py:243-245
    assert resolve_metric(None, "cuvs") == "cosine"            # cu_knn.py:247-248
    assert resolve_metric(None, "torch") == "one_minus_dot"    # cu_knn.py:256-257
    with pytest.raises(ValueError):
        resolve_metric(None, "faiss")
    with pytest.raises(ValueError):
        resolve_metric("manhattan")


def test_tune_memory_reproduces_the_readme_run(monkeypatch):
    """docs/cli.png: 16 GiB card, 100k-row probe of d=1536 float32 = 614 400 000 B, factor 6, threshold 480 000."""
    monkeypatch.setattr(cu_knn, "mem_info", lambda device=0: (0, 17179869184))
    table = pa.table({f"embedding_{i}": pa.array(np.zeros(4, np.float32)) for i in range(1536)})

    class Big:                       # 4M-row table without allocating it
        schema = table.schema
        def __len__(self): return 4_000_000
    assert cu_knn.tune_memory(Big(), 100000, 0.1, None) == 480000
    class Small(Big):
        def __len__(self): return 300_000
    assert cu_knn.tune_memory(Small(), 100000, 0.1, None) == 300_000       # tile >= table -> whole table
    # under the threshold the reference's float batch size trips table.slice and lands on 0.8*1.2 (cu_knn.py:100-107)
    assert cu_knn.tune_memory(Big(), 100000, 0.5, None) == int(0.8 * (600000 * 1.2))


def test_tune_memory_on_a_288gb_card(monkeypatch):
    monkeypatch.setattr(cu_knn, "mem_info", lambda device=0: (0, 288 * 2 ** 30))
    schema = pa.schema([(f"embedding_{i}", pa.float32()) for i in range(1024)])
    class Base:
        def __len__(self): return 10_000_000
    Base.schema = schema
    assert cu_knn.tune_memory(Base(), 100000, 0.1, None) == 10_000_000     # the whole C3 base is one tile


def test_get_file_count_counts_only_numbered_index_files(tmp_path):
    p = tmp_path / "partial"; p.mkdir()
    for name in ["indices0.parquet", "indices1.parquet", "indices12.parquet", "final_indices.parquet", "distances0.parquet",
                 "indicesX.parquet", "indices3.parquet.tmp"]:
        (p / name).write_bytes(b"")
    assert merge.get_file_count(str(p)) == 3                              # merge.py:15-27
    assert merge.get_file_count(str(tmp_path / "nothing")) == 0


def test_stream_to_parquet_contract(tmp_path):
    a = np.arange(250 * 7, dtype=np.float32).reshape(250, 7)
    f = tmp_path / "distances0.parquet"
    cu_knn.stream_cudf_to_parquet(str(f), a, 100)
    md = pq.ParquetFile(f)
    assert md.metadata.num_row_groups == 3 and md.metadata.num_rows == 250   # chunk_size rows per group, cu_knn.py:26-51
    t = pq.read_table(f)
    assert t.schema.names == [str(i) for i in range(7)]                      # cu_knn.py:278-279
    assert all(pa.types.is_float32(x.type) for x in t.schema)
    i = np.arange(20, dtype=np.int64).reshape(4, 5)
    cu_knn.stream_cudf_to_parquet(str(tmp_path / "i.parquet"), i, 100000)
    assert all(pa.types.is_int64(x.type) for x in pq.read_table(tmp_path / "i.parquet").schema)


def test_resume_accepts_a_partial_pair_only_from_the_same_kind_of_run(tmp_path):
    """--resume: a pair counts as finished only with the right rows AND columns AND the metric / dtype / tile bounds it was
    written with (kept in the parquet key-value metadata); untagged files (the reference's, an older build's) do not match."""
    (tmp_path / "partial").mkdir()
    d = np.zeros((30, 5), dtype=np.float32); i = np.zeros((30, 5), dtype=np.int64)
    tag = cu_knn._tile_tag("sqeuclidean", "fp32", 1000, 500)

    def write(t):
        cu_knn.stream_cudf_to_parquet(nw_utils.get_partial_distances_filename(str(tmp_path), 0), d, 100000, t)
        cu_knn.stream_cudf_to_parquet(nw_utils.get_partial_indices_filename(str(tmp_path), 0), i, 100000, t)

    write(tag)
    ok = lambda *a: cu_knn._partial_pair_exists(str(tmp_path), 0, *a)
    assert ok(30) and ok(30, 5) and ok(30, 5, tag)
    assert not ok(31, 5, tag) and not ok(30, 6, tag)
    assert not ok(30, 5, cu_knn._tile_tag("cosine", "fp32", 1000, 500))
    assert not ok(30, 5, cu_knn._tile_tag("sqeuclidean", "fp16", 1000, 500))
    assert not ok(30, 5, cu_knn._tile_tag("sqeuclidean", "fp32", 0, 500))
    assert not cu_knn._partial_pair_exists(str(tmp_path), 1, 30)
    write(None)
    assert ok(30, 5) and not ok(30, 5, tag)
    t = pq.read_table(nw_utils.get_partial_distances_filename(str(tmp_path), 0))       # the tag does not disturb readers
    assert t.schema.names == [str(j) for j in range(5)]


def test_launcher_exits_with_mains_return_code():
    from pathlib import Path
    src = (Path(__file__).resolve().parents[1] / "launcher.py").read_text()
    assert "sys.exit(main())" in src                                                      # launcher.py of the reference discards it


def test_prep_table_and_matrix(tmp_path):
    f = synth.write_source_parquet(str(tmp_path / "m_8_base_vector_data_30.parquet"), 5, 30, 8, chunk_rows=16)
    t = cu_knn.prep_table(str(tmp_path), "m", "m_8_base_vector_data_30.parquet", 20, 8)
    assert t.schema.names == [f"embedding_{i}" for i in range(8)] and len(t) == 20      # cu_knn.py:132-148
    m = cu_knn.table_to_matrix(t)
    assert m.dtype == np.float32 and m.flags["C_CONTIGUOUS"] and np.array_equal(m, synth.gen_rows(5, 0, 20, 8))
    with pytest.raises(AssertionError, match="Expected 9 values, got 8"):
        cu_knn.prep_table(str(tmp_path), "m", "m_8_base_vector_data_30.parquet", 20, 9)  # cu_knn.py:136-137
    full = pq.read_table(f)
    assert full.schema.names[:4] == ["document_id_idx", "text", "title", "document_id"]  # tests/test_02_cu_knn.py:15-22


def test_compute_knn_rejects_a_last_tile_smaller_than_k(tmp_path):
    synth.write_source_parquet(str(tmp_path / "q.parquet"), 1, 4, 8)
    synth.write_source_parquet(str(tmp_path / "b.parquet"), 2, 105, 8)
    with pytest.raises(AssertionError, match="Cannot generate k of 10 with only 105 rows and batch_size of 100"):
        cu_knn.compute_knn(str(tmp_path), "m", 8, "q.parquet", 4, "b.parquet", 105, False, 10, 100)   # cu_knn.py:177-178


def test_shard_bounds_cover_the_base_exactly():
    for n, w in [(10_000_000, 8), (1003, 7), (5, 5), (130, 2)]:
        edges = [shard_bounds(n, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[r][1] == edges[r + 1][0] for r in range(w - 1))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_product_generator_equals_the_oracle_generator(oracle):
    assert np.array_equal(synth.gen_rows(synth.SEED_BASE, 11, 40, 50), oracle.gen_numpy(oracle.SEED_BASE, 11, 40, 50))
    assert np.array_equal(synth.gen_unit_rows(9, 0, 40, 384), oracle.gen_unit_rows(9, 0, 40, 384))
    u = synth.gen_unit_rows(9, 0, 40, 384)
    assert np.abs(np.linalg.norm(u.astype(np.float64), axis=1) - 1).max() < 2e-7        # tests/test_02_cu_knn.py:38-41


def test_cli_keeps_the_reference_flags():
    from neighborhoodwatch_amd.neighborhoodwatch import build_parser
    a = build_parser().parse_args(["1000", "10000", "-k", "100", "-m", "intfloat/e5-small-v2", "--disable-memory-tuning"])
    assert (a.query_count, a.base_count, a.k, a.model_name) == (1000, 10000, 100, "intfloat/e5-small-v2")
    assert a.enable_memory_tuning is False and a.use_dataset_api is False and a.gen_hdf5 is True and a.post_validation is False
    assert a.data_dir == "knn_dataset" and a.output_dtype == "float" and a.output_dimension_size is None
    a = build_parser().parse_args(["1", "2", "-m", "x", "--enable-memory-tuning", "--use-dataset-api", "--no-gen-hdf5", "-ods", "256",
                                   "-odt", "int8", "--data-dir", "/tmp/x", "--post-validation"])
    assert a.enable_memory_tuning and a.use_dataset_api and not a.gen_hdf5 and a.output_dimension_size == 256 and a.post_validation


def test_ck_cli_keeps_the_reference_flags():
    from neighborhoodwatch_amd.colbert_knn import build_parser                       # colbert_knn.py:149-172
    a = build_parser().parse_args(["100000", "1000000", "-k", "100", "--disable-memory-tuning"])
    assert (a.query_token_count, a.base_token_count, a.k, a.model_name, a.engine) == (100000, 1000000, 100, "colbertv2.0", "torch")
    assert a.embedding_scale == "medium" and a.data_dir == "knn_dataset" and a.gen_hdf5 is True and a.use_dataset_api is False
    a = build_parser().parse_args(["1", "2", "-es", "small", "--engine", "raft", "--enable-memory-tuning", "--no-gen-hdf5", "--data-dir", "/tmp/x"])
    assert a.embedding_scale == "small" and a.engine == "raft" and a.enable_memory_tuning and not a.gen_hdf5


def test_engine_refuses_tensors_of_another_gpu():
    """A CUDA tensor handed to an engine must live on the engine's GPU: the check comes before anything touches a stream, so it
    can be exercised without a device (the engine object is made without nwknn_create)."""
    from types import SimpleNamespace
    from neighborhoodwatch_amd.engine import KnnEngine
    eng = KnnEngine.__new__(KnnEngine)
    eng.device = 0
    eng._h = None
    other = SimpleNamespace(device=SimpleNamespace(index=1))
    with pytest.raises(ValueError, match="cuda:0"):
        eng._order_after_torch(other, "queries")


def test_table_columns_hands_over_chunks_without_copying(tmp_path):
    """pq.read_table yields one chunk per parquet row group: table_columns returns the chunks as zero-copy views (a list per
    column), not a concatenation; a single-chunk table gives plain arrays; differently chunked columns are concatenated."""
    f = synth.write_source_parquet(str(tmp_path / "m_8_base_vector_data_50.parquet"), 5, 50, 8, chunk_rows=16)
    t = cu_knn.prep_table(str(tmp_path), "m", "m_8_base_vector_data_50.parquet", 50, 8)
    cols = cu_knn.table_columns(t)
    assert len(cols) == 8 and all(isinstance(c, list) and [len(a) for a in c] == [16, 16, 16, 2] for c in cols)
    ref = synth.gen_rows(5, 0, 50, 8)
    assert all(np.array_equal(np.concatenate(cols[j]), ref[:, j]) for j in range(8))
    assert not cols[0][0].flags["OWNDATA"]                                   # a view of the arrow buffer
    one = cu_knn.table_columns(t.slice(3, 9))                                # inside the first row group: one chunk
    assert all(isinstance(c, np.ndarray) and np.array_equal(c, ref[3:12, j]) for j, c in enumerate(one))
    mixed = pa.table({"embedding_0": pa.chunked_array([[1.0, 2.0], [3.0]], type=pa.float32()),
                      "embedding_1": pa.chunked_array([[4.0], [5.0, 6.0]], type=pa.float32())})
    m = cu_knn.table_columns(mixed)
    assert all(isinstance(c, np.ndarray) for c in m) and m[1].tolist() == [4.0, 5.0, 6.0]


def test_parquet_rows_reads_only_what_a_tile_needs(tmp_path):
    """ParquetRows: a lazy window with a pyarrow.Table's len / schema / slice; a slice decodes the row groups that overlap it and
    nothing else; windows over several files number their rows through."""
    from neighborhoodwatch_amd.parquet_stream import ParquetRows
    f1 = synth.write_source_parquet(str(tmp_path / "a.parquet"), 5, 50, 8, chunk_rows=16)
    f2 = synth.write_source_parquet(str(tmp_path / "b.parquet"), 6, 30, 8, chunk_rows=7)
    ref = np.concatenate([synth.gen_rows(5, 0, 50, 8), synth.gen_rows(6, 0, 30, 8)])
    cols = [f"embedding_{i}" for i in range(8)]
    rows = ParquetRows([f1, f2], 77, cols)
    assert len(rows) == 77 and rows.num_rows == 77 and rows.schema.names == cols and rows.column_names == cols
    for off, ln in [(0, 77), (3, 9), (15, 20), (40, 30), (49, 3), (70, 100), (77, 5), (16, 16)]:
        t = rows.slice(off, ln)
        assert np.array_equal(cu_knn.table_to_matrix(t), ref[off:min(off + ln, 77)]), (off, ln)
    assert [len(t) for t in rows.iter_tables(20)] == [20, 20, 20, 17]
    assert [len(t) for t in rows.iter_tables(20, start=40)] == [20, 17]
    opened = []
    real = pq.ParquetFile.read_row_groups
    try:
        pq.ParquetFile.read_row_groups = lambda self, groups, **kw: (opened.append(list(groups)), real(self, groups, **kw))[1]
        ParquetRows(f1, 50, cols).slice(17, 10)                                  # rows 17..26 live in row group 1 (rows 16..31)
    finally:
        pq.ParquetFile.read_row_groups = real
    assert opened == [[1]]
    assert len(ParquetRows(f1).select(["embedding_3"]).head(5).to_table()) == 5
    with pytest.raises(KeyError):
        ParquetRows(f1, 10, ["no_such_column"])


def test_piece_reader_hands_row_groups_over_in_order_a_bounded_number_ahead(tmp_path):
    """ParquetRows.pieces / PieceReader: the row-group pieces of a row window (clipped at both ends, over two files) decoded by several
    reader threads and handed out IN ROW ORDER, never more than `ahead` beyond the consumer; a reader's exception surfaces at its piece."""
    import time
    from neighborhoodwatch_amd.parquet_stream import ParquetRows, PieceReader
    f1 = synth.write_source_parquet(str(tmp_path / "a.parquet"), 5, 50, 8, chunk_rows=16)
    f2 = synth.write_source_parquet(str(tmp_path / "b.parquet"), 6, 30, 8, chunk_rows=7)
    ref = np.concatenate([synth.gen_rows(5, 0, 50, 8), synth.gen_rows(6, 0, 30, 8)])
    cols = [f"embedding_{i}" for i in range(8)]
    rows = ParquetRows([f1, f2], 77, cols)
    for off, ln in [(0, 77), (3, 9), (15, 20), (40, 30), (49, 3), (70, 100), (16, 16)]:
        ps = rows.pieces(off, ln)
        end = min(off + ln, 77)
        assert ps[0][2] == off and sum(p[4] for p in ps) == end - off and all(a[2] + a[4] == b[2] for a, b in zip(ps[:-1], ps[1:]))
        got = np.concatenate([cu_knn.table_to_matrix(rows.read_piece(p)) for p in ps])
        assert np.array_equal(got, ref[off:end]), (off, ln)
    assert rows.pieces(77, 5) == []
    ps = rows.pieces(0, 77)
    assert len(ps) == 4 + 4                                                     # 50 rows in groups of 16, then 27 of the 30 in groups of 7
    with PieceReader([(rows, p) for p in ps], readers=3, ahead=2) as r:
        out = []
        for t in r:
            time.sleep(0.005)
            out.append(cu_knn.table_to_matrix(t))
        assert np.array_equal(np.concatenate(out), ref[:77]) and r.max_inflight <= 2

    class Broken(ParquetRows):
        def read_piece(self, piece):
            if piece[2] >= 32:
                raise OSError("disk went away")
            return ParquetRows.read_piece(self, piece)
    bad = Broken([f1], 50, cols)
    with PieceReader([(bad, p) for p in bad.pieces(0, 50)], readers=2) as r:
        assert len(next(r)) == 16 and len(next(r)) == 16
        with pytest.raises(OSError, match="disk went away"):
            next(r)
    assert cu_knn.base_geometry(rows.schema) == (8, "f32")


def test_prefetch_keeps_two_items_alive_and_hands_errors_over():
    import time
    from neighborhoodwatch_amd.parquet_stream import Prefetch

    def gen():
        for i in range(8):
            yield i

    p = Prefetch(gen, slots=2)
    got = []
    for x in p:
        time.sleep(0.01)                                                         # the producer runs ahead while the consumer works
        got.append(x)
    assert got == list(range(8)) and p.max_alive == 2

    def bad():
        yield 1
        raise RuntimeError("boom")

    with pytest.raises(RuntimeError, match="boom"):
        for _ in Prefetch(bad, slots=2):
            pass
    with Prefetch(gen, slots=1) as q:                                            # closing early stops the producer
        assert next(q) == 0


def test_slice_dataset_is_lazy_and_checkpoints_round_trip(tmp_path):
    from neighborhoodwatch_amd import cu_knn_ds
    from neighborhoodwatch_amd.parquet_stream import ParquetRows
    f = synth.write_source_parquet(str(tmp_path / "b.parquet"), 6, 30, 8, chunk_rows=7)
    d = cu_knn_ds.load_dataset(str(tmp_path), "b.parquet")
    w = cu_knn_ds.slice_dataset(d, 0, 20)                                        # cu_knn_ds.py:48-49, without materialising
    assert isinstance(w, ParquetRows) and cu_knn_ds.dataset_count(w) == 20 and cu_knn_ds.dataset_count(d) == 30
    assert cu_knn_ds.get_dataset_columns(w, 8)[-1] == "embedding_7"
    (tmp_path / "partial").mkdir()
    rng = np.random.default_rng(1)
    dist = np.sort(rng.random((12, 5)).astype(np.float32), axis=1); idx = rng.integers(0, 1000, (12, 5)).astype(np.int64)
    cu_knn_ds._write_checkpoint(str(tmp_path), dist, idx, cu_knn_ds._checkpoint_tag("sqeuclidean", 5, 12, 1000, 400))
    ck = cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 5, 12, 1000)
    assert ck[0] == 400 and np.array_equal(ck[1], dist) and np.array_equal(ck[2], idx)
    assert cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 6, 12, 1000) is None       # other k
    assert cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 5, 12, 999) is None        # other base
    assert merge.get_file_count(str(tmp_path / "partial")) == 0                             # checkpoints are not partial pairs
    cu_knn_ds._drop_checkpoint(str(tmp_path))
    assert cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 5, 12, 1000) is None
    # a checkpoint is tied to its INPUT FILES (path, size, mtime, columns), not only to counts: another job's leftover with the same
    # shape in the same data_dir is not resumed
    rows = cu_knn_ds._as_rows(w, cu_knn_ds.get_dataset_columns(w, 8))
    prov = rows.provenance()
    assert prov == cu_knn_ds._as_rows(cu_knn_ds.slice_dataset(d, 0, 20), cu_knn_ds.get_dataset_columns(w, 8)).provenance()
    assert prov != cu_knn_ds._as_rows(cu_knn_ds.slice_dataset(d, 0, 19), cu_knn_ds.get_dataset_columns(w, 8)).provenance()
    cu_knn_ds._write_checkpoint(str(tmp_path), dist, idx, cu_knn_ds._checkpoint_tag("sqeuclidean", 5, 12, 1000, 400, f"q:x;b:{prov}"))
    assert cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 5, 12, 1000, f"q:x;b:{prov}")[0] == 400
    import os
    os.utime(f, ns=(1, 1))                                                       # the base file changes under the same name and size
    prov2 = cu_knn_ds._as_rows(cu_knn_ds.slice_dataset(d, 0, 20), cu_knn_ds.get_dataset_columns(w, 8)).provenance()
    assert prov2 != prov and cu_knn_ds._load_checkpoint(str(tmp_path), "sqeuclidean", 5, 12, 1000, f"q:x;b:{prov2}") is None
    # the materialising fallback of slice_dataset (start > 0) yields a dataset with no files behind it: rows come from memory
    from neighborhoodwatch_amd.parquet_stream import TableRows
    mem = cu_knn_ds.slice_dataset(d, 5, 25)
    mrows = cu_knn_ds._as_rows(mem, cu_knn_ds.get_dataset_columns(mem, 8))
    assert isinstance(mrows, TableRows) and len(mrows) == 20 and mrows.column_names == [f"embedding_{j}" for j in range(8)]
    assert mrows.slice(3, 4).num_rows == 4 and mrows.provenance().startswith("memory:")
    assert np.array_equal(mrows.to_table().column("embedding_0").to_numpy(), synth.gen_rows(6, 5, 20, 8)[:, 0])
    with pytest.raises(TypeError):
        cu_knn_ds._as_rows(object(), ["embedding_0"])


def test_fixed_size_list_embedding_column(tmp_path):
    """SURVEY N2's optional layout: ONE FixedSizeList<float32, d> column instead of d float columns.  prep_table validates d from the
    list size, keeps only that column (eager or lazy), table_input hands the engine the row-major values -- a zero-copy view for a
    one-chunk tile, one concatenation across row groups -- and tune_memory's footprint counts d floats per row."""
    f = synth.write_source_parquet(str(tmp_path / "m_8_base_vector_data_50.parquet"), 5, 50, 8, chunk_rows=16, layout="list")
    assert pq.read_schema(f).names[-1] == "embedding" and pa.types.is_fixed_size_list(pq.read_schema(f).field("embedding").type)
    ref = synth.gen_rows(5, 0, 50, 8)
    for lazy in (False, True):
        t = cu_knn.prep_table(str(tmp_path), "m", "m_8_base_vector_data_50.parquet", 45, 8, lazy=lazy)
        assert len(t) == 45 and list(t.schema.names) == ["embedding"] and cu_knn.float_row_bytes(t.schema) == 32
        kind, m = cu_knn.table_input(t.slice(10, 30))
        assert kind == "matrix" and m.dtype == np.float32 and np.array_equal(m, ref[10:40])
        kind, m1 = cu_knn.table_input(t.slice(17, 9))                              # inside one row group
        assert np.array_equal(m1, ref[17:26])
    with pytest.raises(AssertionError, match="Expected 9 values, got 8"):
        cu_knn.prep_table(str(tmp_path), "m", "m_8_base_vector_data_50.parquet", 45, 9)
    w = cu_knn_ds.slice_dataset(cu_knn_ds.load_dataset(str(tmp_path), "m_8_base_vector_data_50.parquet"), 0, 40)
    assert cu_knn_ds.get_dataset_columns(w, 8) == ["text", "embedding"]
    kind, cols = cu_knn.table_input(pq.read_table(synth.write_source_parquet(str(tmp_path / "c.parquet"), 5, 20, 8)).select(
        [f"embedding_{i}" for i in range(8)]))
    assert kind == "columns" and len(cols) == 8                                    # the reference's layout still takes the column path


@pytest.mark.parametrize("env", [{"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"}, {"ROCPROF_OUTPUT_PATH": "/tmp/x"},
                                 {"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofv3-list-avail.so"}])
def test_bench_does_not_start_ranks_from_under_a_profiler(env):
    """`rocprofv3 ... -- python3 bench.py --gpus N`: the profiler's preloaded library has initialised the GPU before main(), so starting
    the N ranks from that process is the exec-after-GPU-init the pool forbids.  bench.py sees the preload and refuses (exit 4) before
    anything is started; without the preload it would go on to start the ranks (not done here: no GPU)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    import bench
    old = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update(env)
        assert bench.profiler_preload()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    assert bench.profiler_preload() == ""
    if "LD_PRELOAD" in env:
        return                                     # (a missing preload library would only add loader noise to the child below)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--workload", "dev"], env={**os.environ, **env},
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 4 and "refusing to start 2 ranks from under a profiler" in r.stderr and r.stdout == ""


def test_effective_cpus_honours_the_cgroup_quota_and_sizes_pyarrows_pool(monkeypatch, tmp_path):
    """hostinfo.effective_cpus: the affinity mask capped by the cgroup CPU quota (cgroup v2 ``cpu.max``: "<quota> <period>" or "max <period>"),
    and size_arrow_pool only ever SHRINKS pyarrow's CPU pool (NWKNN_ARROW_THREADS sets it outright, 0 leaves it alone)."""
    import builtins
    import pyarrow as pa
    from neighborhoodwatch_amd import hostinfo
    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **kw):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                if content is None:
                    raise FileNotFoundError(path)
                f = tmp_path / "cpu.max"
                f.write_text(content)
                return real_open(f, *a, **kw)
            if str(path).startswith("/sys/fs/cgroup/cpu/"):
                raise FileNotFoundError(path)
            return real_open(path, *a, **kw)
        return _open

    monkeypatch.setattr(hostinfo.os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000"))
    assert hostinfo.effective_cpus() == 16
    monkeypatch.setattr(builtins, "open", fake("150000 100000"))
    assert hostinfo.effective_cpus() == 2                               # a quota of 1.5 CPUs: rounded up
    monkeypatch.setattr(builtins, "open", fake("max 100000"))
    assert hostinfo.effective_cpus() == 256
    monkeypatch.setattr(builtins, "open", fake(None))
    assert hostinfo.effective_cpus() == 256

    before = pa.cpu_count()
    try:
        monkeypatch.setattr(builtins, "open", fake("200000 100000"))
        pa.set_cpu_count(8)
        monkeypatch.setattr(hostinfo, "_arrow_sized", False)
        monkeypatch.delenv("NWKNN_ARROW_THREADS", raising=False)
        assert hostinfo.size_arrow_pool() == 2                          # shrunk to the quota
        pa.set_cpu_count(1)
        monkeypatch.setattr(hostinfo, "_arrow_sized", False)
        assert hostinfo.size_arrow_pool() == 1                          # never grown
        monkeypatch.setattr(hostinfo, "_arrow_sized", False)
        monkeypatch.setenv("NWKNN_ARROW_THREADS", "3")
        assert hostinfo.size_arrow_pool() == 3                          # set outright
        pa.set_cpu_count(5)
        monkeypatch.setattr(hostinfo, "_arrow_sized", False)
        monkeypatch.setenv("NWKNN_ARROW_THREADS", "0")
        assert hostinfo.size_arrow_pool() == 5                          # left alone
    finally:
        pa.set_cpu_count(before)


def test_bench_progress_lines_go_to_stderr_of_rank_0_only(capsys, monkeypatch):
    """bench.py's stdout is ONE JSON line (the driver parses it); what a long run says while it works -- warm-ups done, a step count every
    ~45 s, every secondary leg -- goes to stderr, from rank 0 only.  (A default run is ~425 s long: silent, it was taken for a hang.)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    monkeypatch.delenv("RANK", raising=False)
    bench.progress("c3: step 4 of 20")
    out = capsys.readouterr()
    assert out.out == "" and out.err.startswith("bench.py [") and out.err.endswith("c3: step 4 of 20\n")
    monkeypatch.setenv("RANK", "3")
    bench.progress("never printed")
    out = capsys.readouterr()
    assert out.out == "" and out.err == ""
