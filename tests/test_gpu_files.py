"""GPU: the on-disk contract end to end -- source parquet -> compute_knn / compute_knn_ds -> partial parquet ->
merge_indices_and_distances -> final parquet, compared with the oracle on the whole base."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from neighborhoodwatch_amd import cu_knn, cu_knn_ds, merge, nw_utils, synth

pytestmark = pytest.mark.gpu


def _final(data_dir):
    fi = pq.read_table(nw_utils.get_partial_indices_filename(data_dir, -1))
    fd = pq.read_table(nw_utils.get_partial_distances_filename(data_dir, -1))
    return fd, fi


def _arr(t, dtype):
    return np.stack([t.column(j).to_numpy() for j in range(t.num_columns)], axis=1).astype(dtype)


@pytest.fixture()
def dataset(tmp_path):
    data_dir = nw_utils.setup_model_output_folder(str(tmp_path), "intfloat/e5-small-v2", 300, 2600, 10)
    qf = nw_utils.get_source_query_dataset_filename(data_dir, "intfloat/e5-small-v2", 300, 384)
    bf = nw_utils.get_source_base_dataset_filename(data_dir, "intfloat/e5-small-v2", 2600, 384)
    synth.write_source_parquet(qf, 0x71, 300, 384, normalize=True)
    synth.write_source_parquet(bf, 0x72, 2600, 384, normalize=True, chunk_rows=700)
    return data_dir, qf, bf


def test_compute_knn_then_merge_equals_oracle(dataset, oracle):
    data_dir, qf, bf = dataset
    # 2600 base rows in tiles of 1000 -> 3 partial pairs (1000, 1000, 600), cu_knn.py:176,201-205
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 1000)
    assert merge.get_file_count(f"{data_dir}/partial") == 3
    p1 = pq.read_table(nw_utils.get_partial_indices_filename(data_dir, 1))
    assert p1.schema.names == [str(c) for c in range(10)] and len(p1) == 300             # cu_knn.py:278-282
    assert all(pa.types.is_int64(f.type) for f in p1.schema)
    ids1 = _arr(p1, np.int64)
    assert ids1.min() >= 1000 and ids1.max() < 2000                                       # + batch_offset, cu_knn.py:273
    pd1 = pq.read_table(nw_utils.get_partial_distances_filename(data_dir, 1))
    assert all(pa.types.is_float32(f.type) for f in pd1.schema)

    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    assert fd.schema.names == fi.schema.names == [str(c) for c in range(10)]
    assert all(pa.types.is_float32(f.type) for f in fd.schema) and all(pa.types.is_int64(f.type) for f in fi.schema)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri)
    assert np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))
    # the reference's own validator: isclose(1 - dot(q, b), distance / 2)   (parquet_to_format.py:368-373)
    d = _arr(fd, np.float32); i = _arr(fi, np.int64)
    for r in range(0, 300, 29):
        assert np.isclose(1.0 - base[i[r]].astype(np.float64) @ queries[r].astype(np.float64), d[r] / 2.0).all()


def test_single_tile_merge_is_a_pass_through(dataset, oracle):
    data_dir, qf, bf = dataset
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 100000, engine="torch")
    assert merge.get_file_count(f"{data_dir}/partial") == 1
    merge.merge_indices_and_distances(data_dir, k=10)                                     # merge.py:67-69
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10, "one_minus_dot")
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_row_counts_slice_the_files(dataset, oracle):
    data_dir, qf, bf = dataset
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 120, bf, 1500, False, 10, 500, engine="cuvs")   # .slice(0,count), cu_knn.py:114
    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 1500, 384); queries = synth.gen_unit_rows(0x71, 0, 120, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10, "cosine")
    assert len(fi) == 120 and np.array_equal(_arr(fi, np.int64), ri)
    assert np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


@pytest.mark.parametrize("prescreen", [False, True])
def test_dataset_api_writes_the_final_tables_once(dataset, oracle, prescreen):
    data_dir, qf, bf = dataset
    cu_knn_ds.compute_knn_ds(data_dir, 384, qf, 300, bf, 2600, False, 10, 800, prescreen=prescreen)   # 4 base batches: 800,800,800,200
    assert merge.get_file_count(f"{data_dir}/partial") == 0
    merge.merge_indices_and_distances(data_dir, k=10)                                     # no partial files: no-op, merge.py:40-42
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_dataset_api_streams_batches_and_resumes_from_its_checkpoint(dataset, oracle, monkeypatch):
    """7 base batches of 400 rows (the last one 200: unequal), row groups of 700 rows so batches straddle them.  The first run dies
    inside its fourth batch; the second, with resume=True, picks up the checkpoint written after the third batch (1 200 rows done),
    searches only the remaining four and ends in the oracle's bits.  Host memory held at most two batches at any time; the running
    pair stayed on the device between batches."""
    from neighborhoodwatch_amd import engine as eng_mod
    from neighborhoodwatch_amd import parquet_stream
    data_dir, qf, bf = dataset
    # every batch's search is attached to the batch before its first piece (search_attach: the global offset goes in there) and collected
    # once the batch is resident (search_collect: results asked for on the device)
    real_attach, real_collect = eng_mod.KnnEngine.search_attach, eng_mod.KnnEngine.search_collect
    calls = []

    def attach(self, queries, k, metric="sqeuclidean", index_offset=0):
        calls.append([index_offset, None])
        ok = real_attach(self, queries, k, metric, index_offset)
        assert ok
        return ok

    def flaky(self, out_device=False):
        calls[-1][1] = out_device
        if len(calls) == 4:
            raise RuntimeError("simulated crash in batch 4")
        return real_collect(self, out_device)

    monkeypatch.setattr(eng_mod.KnnEngine, "search_attach", attach)
    monkeypatch.setattr(eng_mod.KnnEngine, "search_collect", flaky)
    with pytest.raises(RuntimeError, match="simulated crash"):
        cu_knn_ds.compute_knn_ds(data_dir, 384, qf, 300, bf, 2600, False, 10, 400, resume=True)
    assert [c[0] for c in calls] == [0, 400, 800, 1200] and all(c[1] for c in calls)     # global offsets; results asked for on the device
    def rows_of(name, count):
        ds_ = cu_knn_ds.slice_dataset(cu_knn_ds.load_dataset(data_dir, name), 0, count)
        return cu_knn_ds._as_rows(ds_, cu_knn_ds.get_dataset_columns(ds_, 384))
    inputs = cu_knn_ds.checkpoint_inputs(rows_of(qf, 300), rows_of(bf, 2600))
    ck = cu_knn_ds._load_checkpoint(data_dir, "sqeuclidean", 10, 300, 2600, inputs)
    assert ck is not None and ck[0] == 1200
    assert cu_knn_ds._load_checkpoint(data_dir, "cosine", 10, 300, 2600, inputs) is None  # another metric's checkpoint is not trusted
    assert cu_knn_ds._load_checkpoint(data_dir, "sqeuclidean", 10, 300, 2600) is None     # nor one whose input files are not these
    assert cu_knn_ds._load_checkpoint(data_dir, "sqeuclidean", 10, 300, 2600, cu_knn_ds.checkpoint_inputs(rows_of(qf, 300), rows_of(bf, 2599))) is None

    monkeypatch.setattr(eng_mod.KnnEngine, "search_collect", real_collect)
    seen = []
    real_reader = parquet_stream.PieceReader

    class Watched(real_reader):
        def __init__(self, pieces, **kw):
            super().__init__(pieces, ahead=2, **kw)                                              # the bound under test (the default is sized by bytes)
            seen.append(self)

    monkeypatch.setattr(cu_knn, "PieceReader", Watched)                                         # cu_knn.stream_tiles feeds both drivers
    calls2 = []
    monkeypatch.setattr(eng_mod.KnnEngine, "search_attach",
                        lambda self, queries, k, metric="sqeuclidean", index_offset=0:
                        (calls2.append(index_offset), real_attach(self, queries, k, metric, index_offset))[1])
    cu_knn_ds.compute_knn_ds(data_dir, 384, qf, 300, bf, 2600, False, 10, 400, resume=True)
    assert calls2 == [1200, 1600, 2000, 2400]                                             # the first three batches were not searched again
    # ONE reader pool for the whole job: the row-group pieces of the remaining batches in row order (rows 1200..2599 over 700-row groups:
    # 1200-1399 | 1400-1599, 1600-1999, 2000-2099 | 2100-2399, 2400-2599, cut at batch borders too), never more than two decoded ahead
    assert len(seen) == 1 and seen[0].max_inflight <= 2 and len(seen[0]._pieces) == 6
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))
    import os
    assert not any(os.path.exists(f) for f in cu_knn_ds._checkpoint_names(data_dir))      # a finished job leaves no checkpoint behind


def test_compute_knn_reads_tiles_lazily_and_overlaps(dataset, oracle, monkeypatch):
    """compute_knn opens the base file lazily and hands every tile to the GPU PIECE BY PIECE: the row groups that overlap a tile are decoded
    by reader threads (each piece read once, nothing up front, a bounded number ahead) and appended to the engine's base as they arrive --
    nwknn_base_begin / _append_column_chunks / _end -- while the previous tile is searched; the partial pairs -- written by a writer
    thread -- are the reference layout."""
    from neighborhoodwatch_amd import engine as eng_mod
    from neighborhoodwatch_amd import parquet_stream
    data_dir, qf, bf = dataset
    seen = []
    real_reader = parquet_stream.PieceReader

    class Watched(real_reader):
        def __init__(self, pieces, **kw):
            super().__init__(pieces, ahead=2, **kw)
            seen.append(self)

    monkeypatch.setattr(cu_knn, "PieceReader", Watched)
    read_calls, begun, appended = [], [], []
    real_read = parquet_stream.ParquetRows.read_piece
    monkeypatch.setattr(parquet_stream.ParquetRows, "read_piece", lambda self, pc: (read_calls.append((pc[2], pc[4])), real_read(self, pc))[1])
    real_begin, real_append = eng_mod.KnnEngine.base_begin, eng_mod.KnnEngine.base_append_columns
    monkeypatch.setattr(eng_mod.KnnEngine, "base_begin", lambda self, rows, dim, dtype="f32": (begun.append(rows), real_begin(self, rows, dim, dtype))[1])
    # (the pieces arrive prepared -- KnnEngine.prepare_columns on the reader thread: slot 4 of the tuple is the piece's row count)
    monkeypatch.setattr(eng_mod.KnnEngine, "base_append_columns", lambda self, cols: (appended.append(cols[4]), real_append(self, cols))[1])
    attached = []
    real_attach = eng_mod.KnnEngine.search_attach

    def attach(self, queries, k, metric="sqeuclidean", index_offset=0):
        attached.append((index_offset, len(appended), real_attach(self, queries, k, metric, index_offset)))
        return attached[-1][2]

    monkeypatch.setattr(eng_mod.KnnEngine, "search_attach", attach)
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 900)
    # every tile's search was attached before the tile's first piece (global offset, pieces appended so far): knn() walks the tile as it arrives
    assert attached == [(0, 0, True), (900, 2, True), (1800, 4, True)]
    # tiles of 900 / 900 / 800 rows over 700-row groups: (first row, rows) of every piece, each decoded exactly once, in row order
    assert sorted(read_calls) == [(0, 700), (700, 200), (900, 500), (1400, 400), (1800, 300), (2100, 500)]
    assert begun == [900, 900, 800] and appended == [700, 200, 500, 400, 300, 500]
    assert len(seen) == 1 and seen[0].max_inflight <= 2
    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


@pytest.mark.parametrize("api", ["table", "dataset"])
def test_fixed_size_list_layout_end_to_end(tmp_path, oracle, api):
    """Source files holding ONE FixedSizeList<float32, 384> column through both drivers: same partial / final files, the
    oracle's bits."""
    data_dir = nw_utils.setup_model_output_folder(str(tmp_path), "intfloat/e5-small-v2", 150, 2100, 10)
    qf = nw_utils.get_source_query_dataset_filename(data_dir, "intfloat/e5-small-v2", 150, 384)
    bf = nw_utils.get_source_base_dataset_filename(data_dir, "intfloat/e5-small-v2", 2100, 384)
    synth.write_source_parquet(qf, 0x81, 150, 384, normalize=True, layout="list")
    synth.write_source_parquet(bf, 0x82, 2100, 384, normalize=True, chunk_rows=600, layout="list")
    if api == "table":
        cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 150, bf, 2100, False, 10, 800)     # tiles 800, 800, 500
        assert merge.get_file_count(f"{data_dir}/partial") == 3
        merge.merge_indices_and_distances(data_dir, k=10)
    else:
        cu_knn_ds.compute_knn_ds(data_dir, 384, qf, 150, bf, 2100, False, 10, 800)
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x82, 0, 2100, 384); queries = synth.gen_unit_rows(0x81, 0, 150, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_merge_drops_rownum_and_handles_many_lists(tmp_path, oracle):
    """tests/test_03_merge.py:14-24 writes a RowNum column next to the k columns."""
    import pandas as pd
    rng = np.random.default_rng(8)
    n, q, k = 50, 33, 100                              # 50*100 = 5000 entries per query: folded in groups on the GPU
    (tmp_path / "partial").mkdir()
    ds, is_ = [], []
    for l in range(n):
        d = np.sort(rng.random((q, k)).astype(np.float32), axis=1)
        i = rng.integers(0, 10 ** 9, size=(q, k)).astype(np.int64)
        ds.append(d); is_.append(i)
        ddf = pd.DataFrame(d, columns=[str(c) for c in range(k)]); ddf["RowNum"] = np.arange(q)
        idf = pd.DataFrame(i, columns=[str(c) for c in range(k)]); idf["RowNum"] = np.arange(q)
        ddf.to_parquet(tmp_path / "partial" / f"distances{l}.parquet"); idf.to_parquet(tmp_path / "partial" / f"indices{l}.parquet")
    merge.merge_indices_and_distances(str(tmp_path), k=k)
    fd, fi = _final(str(tmp_path))
    assert fd.schema.names == [str(c) for c in range(k)]
    rd, ri = oracle.merge_partials(np.stack(ds), np.stack(is_), k)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


@pytest.mark.parametrize("extra", [[], ["--prescreen"], ["--embedding-layout", "list", "--use-dataset-api"]])
def test_cli_end_to_end(tmp_path, oracle, extra):
    from neighborhoodwatch_amd.neighborhoodwatch import main
    assert main(["200", "3000", "-k", "20", "-m", "intfloat/e5-small-v2", "--data-dir", str(tmp_path), "--disable-memory-tuning"] + extra) == 0
    data_dir = nw_utils.get_model_data_homedir(str(tmp_path), "intfloat/e5-small-v2", 200, 3000, 20)
    fd, fi = _final(data_dir)
    base = synth.gen_rows(synth.SEED_BASE, 0, 3000, 384); queries = synth.gen_rows(synth.SEED_QUERY, 0, 200, 384)
    rd, ri = oracle.knn_canonical(base, queries, 20)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))
    # stage 5: the four fvec/ivec files of neighborhoodwatch.py:167-179
    from neighborhoodwatch_amd import parquet_to_format as ptf
    qv, bv, iv, dv = ptf.get_ivec_fvec_filenames(data_dir, "intfloat_e5-small-v2", 384, 3000, 200, 20)
    assert ptf.count_vectors(data_dir, qv) == 200 and ptf.count_vectors(data_dir, bv) == 3000
    assert np.array_equal(ptf.read_vectors(data_dir, iv), ri.astype(np.int32)) and np.array_equal(ptf.read_vectors(data_dir, dv), rd)


@pytest.mark.parametrize("prescreen", [False, True])
def test_ck_token_level_knn(tmp_path, oracle, prescreen):
    """colbert_knn.process_knn_computation: token_embedding_<i> columns, d=128, 1 - dot, 9000-row base tiles, split=False."""
    from neighborhoodwatch_amd.colbert_knn import process_knn_computation
    (tmp_path / "partial").mkdir()
    synth.write_source_parquet(str(tmp_path / "q_src.parquet"), 0x31, 150, 128, normalize=True, column_prefix="token_embedding_")
    synth.write_source_parquet(str(tmp_path / "b_src.parquet"), 0x32, 20000, 128, normalize=True, column_prefix="token_embedding_")
    # only the vector columns reach the engine (the reference selects float columns, cu_knn.py:208)
    process_knn_computation(str(tmp_path), str(tmp_path / "b_src.parquet"), 20000, str(tmp_path / "q_src.parquet"), 150, k=10,
                            prescreen=prescreen)
    assert merge.get_file_count(f"{tmp_path}/partial") == 3                     # ceil(20000 / 9000), colbert_knn.py:95-98
    merge.merge_indices_and_distances(str(tmp_path), k=10)
    fd, fi = _final(str(tmp_path))
    base = synth.gen_unit_rows(0x32, 0, 20000, 128); queries = synth.gen_unit_rows(0x31, 0, 150, 128)
    rd, ri = oracle.knn_canonical(base, queries, 10, "one_minus_dot")
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_resume_skips_finished_tiles_and_two_engines_share_the_work(dataset, oracle, capsys):
    data_dir, qf, bf = dataset
    # two host threads / two engines (here both on GPU 0) deal the 3 base tiles round-robin
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 1000, devices=[0, 0])
    assert merge.get_file_count(f"{data_dir}/partial") == 3
    import os
    os.remove(nw_utils.get_partial_indices_filename(data_dir, 1))              # tile 1 is "unfinished"
    capsys.readouterr()
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 1000, resume=True)
    out = capsys.readouterr().out
    assert "base tile 0: partial files of the same metric/dtype/tile present, skipped" in out and "base tile 2: partial files of the same" in out
    assert "base tile 1: partial" not in out
    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_resume_does_not_reuse_tiles_of_another_metric(dataset, oracle, capsys):
    """The directory name pins q, b and k but not the distance: partial pairs written by a sqeuclidean run must not be
    mixed into a cosine run by --resume (they carry the metric / dtype / tile bounds in their parquet metadata)."""
    data_dir, qf, bf = dataset
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 1000)            # raft: sqeuclidean
    capsys.readouterr()
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 1000, resume=True, engine="cuvs")
    assert "skipped" not in capsys.readouterr().out
    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    base = synth.gen_unit_rows(0x72, 0, 2600, 384); queries = synth.gen_unit_rows(0x71, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 10, "cosine")
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))


def test_fp16_file_path(dataset, oracle):
    data_dir, qf, bf = dataset
    cu_knn.compute_knn(data_dir, "intfloat/e5-small-v2", 384, qf, 300, bf, 2600, False, 10, 100000, dtype="fp16")
    merge.merge_indices_and_distances(data_dir, k=10)
    fd, fi = _final(data_dir)
    b16 = synth.gen_unit_rows(0x72, 0, 2600, 384).astype(np.float16).astype(np.float32)
    q16 = synth.gen_unit_rows(0x71, 0, 300, 384).astype(np.float16).astype(np.float32)
    ed, ei = oracle.knn_exact_fp64(b16, q16, 10)
    oracle.check_against_exact(b16, q16, _arr(fd, np.float32), _arr(fi, np.int64), ed)


@pytest.mark.parametrize("extra", [[], ["--prescreen"]])
def test_cli_unit_rows_pass_the_reference_validator(tmp_path, extra):
    """nw --normalize --post-validation: stage 5's validate_files_v0 (parquet_to_format.py:351-383: 1 - q.b == distance / 2 for every
    returned neighbour of unit-norm rows) accepts the engine's output, direct and pre-screened."""
    from neighborhoodwatch_amd.neighborhoodwatch import main
    assert main(["150", "4000", "-k", "10", "-m", "intfloat/e5-small-v2", "--data-dir", str(tmp_path), "--disable-memory-tuning",
                 "--normalize", "--post-validation", "--no-gen-hdf5"] + extra) == 0


def test_ck_cli_end_to_end(tmp_path, oracle):
    """`ck` with synthetic token embeddings: same flags and stage order as the reference (colbert_knn.py:146-341), token
    source files with token_embedding_<i> columns only, 9000-row tiles, 1 - dot, merge, fvec/ivec -- checked against the oracle."""
    from neighborhoodwatch_amd import parquet_to_format as p2f
    from neighborhoodwatch_amd.colbert_knn import main
    assert main(["140", "21000", "-k", "10", "--data-dir", str(tmp_path), "--no-gen-hdf5", "--disable-memory-tuning", "--synthetic-seed", "0x40"]) == 0
    data_dir = f"{tmp_path}/colbertv2.0/q140_b21000_k10"
    assert merge.get_file_count(f"{data_dir}/partial") == 3                     # engine torch: 9000-row base tiles (colbert_knn.py:95-98)
    base = synth.gen_unit_rows(0x40, 0, 21000, 128); queries = synth.gen_unit_rows(0x41, 0, 140, 128)
    rd, ri = oracle.knn_canonical(base, queries, 10, "one_minus_dot")
    fd, fi = _final(data_dir)
    assert np.array_equal(_arr(fi, np.int64), ri) and np.array_equal(_arr(fd, np.float32).view(np.uint32), rd.view(np.uint32))
    qf, bf, iif, df_ = p2f.get_ivec_fvec_filenames(data_dir, "colbertv2.0", 128, 21000, 140, 10, None)
    assert p2f.count_vectors(data_dir, iif) == 140 and p2f.count_vectors(data_dir, bf) == 21000


def test_validator_on_the_gpu_gives_the_host_verdict(tmp_path, oracle):
    """validate_files_v0 (parquet_to_format.py:351-383) with the Q x k gather + dot on the GPU (nwknn_pair_distances): the mismatch count
    and the reported pairs are those of the float64 host check -- a clean result, a corrupted distance, a corrupted index and a pair too
    close to the tolerance to be called from the fp32 value (handed to the host)."""
    from neighborhoodwatch_amd import parquet_to_format as ptf
    data_dir = str(tmp_path)
    base, queries = synth.gen_unit_rows(0x51, 0, 20000, 384), synth.gen_unit_rows(0x52, 0, 600, 384)
    queries[17] = 0.0                                                      # the reference skips zero query vectors
    d, i = oracle.knn_canonical(base, queries, 100)
    for name, arr, t in (("q.fvec", queries, "f"), ("b.fvec", base, "f"), ("i.ivec", i.astype(np.int32), "i"), ("d.fvec", d, "f")):
        ptf.write_ivec_fvec_from_dataframe(data_dir, "m", f"{data_dir}/{name}", arr, t, arr.shape[1])
    files = [f"{data_dir}/{n}" for n in ("q.fvec", "b.fvec", "i.ivec", "d.fvec")]
    assert ptf.validate_files_v0(data_dir, *files, device=0) == 0 == ptf.validate_files_v0(data_dir, *files, device=-1)
    bad_d, bad_i = d.copy(), i.astype(np.int32).copy()
    bad_d[5, 2] += 0.01                                                    # a wrong distance
    bad_d[9, 50] *= 1.0 + 0.9e-5                                           # inside numpy.isclose's tolerance, but too close to call in fp32
    bad_d[11, 7] *= 1.0 + 3e-5                                             # outside it
    bad_i[300, 99] = (bad_i[300, 99] + 1234) % 20000                       # a wrong neighbour
    ptf.write_ivec_fvec_from_dataframe(data_dir, "m", files[3], bad_d, "f", 100)
    ptf.write_ivec_fvec_from_dataframe(data_dir, "m", files[2], bad_i, "i", 100)
    host = ptf.validate_files_v0(data_dir, *files, device=-1)
    assert ptf.validate_files_v0(data_dir, *files, device=0) == host and host in (3, 4)
