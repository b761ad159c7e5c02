"""Where the streamed file path spends its time: the refcli job (10k x B x 1536, one float32 column per dimension, 100 000-row groups, /dev/shm)
through cu_knn.compute_knn with the piece loop instrumented -- seconds the consumer waited for a decoded piece (decode-bound), seconds inside
base_append_columns (gather into pinned memory + waiting for a free slab: upload-bound), base_end, search, and the wall clock.
usage: stream_times.py [base_rows] [readers ...]"""
import contextlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyarrow as pa, pyarrow.parquet as pq, torch
from neighborhoodwatch_amd import KnnEngine, cu_knn, engine as E, parquet_stream as PS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
readers_list = [int(x) for x in sys.argv[2:]] or [4]
import cProfile, pstats
d, Q, k = 1536, 10_000, 100
tmp = tempfile.mkdtemp(prefix="nwk_st_", dir="/dev/shm")
try:
    os.makedirs(os.path.join(tmp, "partial"))
    names = [f"embedding_{j}" for j in range(d)]
    with KnnEngine(0) as eng:
        for fname, seed, rows in (("base.parquet", 1, B), ("query.parquet", 2, Q)):
            w = None
            for r0 in range(0, rows, 100_000):
                n = min(100_000, rows - r0)
                cols = eng.generate(seed, r0, n, d).t().contiguous().cpu().numpy()
                t = pa.Table.from_arrays([pa.array(cols[j]) for j in range(d)], names=names)
                if w is None:
                    w = pq.ParquetWriter(os.path.join(tmp, fname), t.schema, compression="NONE", use_dictionary=False, write_statistics=False)
                w.write_table(t)
            w.close()
    acc = {}
    def timed(name, fn):
        def call(*a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return call
    PS.PieceReader.__next__ = timed("wait for a decoded piece", PS.PieceReader.__next__)
    E.KnnEngine.base_append_columns = timed("base_append_columns (gather + slab waits)", E.KnnEngine.base_append_columns)
    E.KnnEngine.base_end = timed("base_end (pad, checks, fp16 copy)", E.KnnEngine.base_end)
    E.KnnEngine.base_begin = timed("base_begin (allocation)", E.KnnEngine.base_begin)
    E.KnnEngine.search_columns = timed("search", E.KnnEngine.search_columns)
    cu_knn.table_input = timed("table_input (arrow -> numpy views)", cu_knn.table_input)
    cu_knn.prep_table = timed("prep_table x2 (query read, base metadata)", cu_knn.prep_table)
    cu_knn.stream_cudf_to_parquet = timed("partial files", cu_knn.stream_cudf_to_parquet)
    PS.PieceReader.__init__ = timed("PieceReader init", PS.PieceReader.__init__)
    PS.PieceReader.close = timed("PieceReader close", PS.PieceReader.close)
    cu_knn.tune_memory = timed("tune_memory", cu_knn.tune_memory)
    for pre in (False, True):
        for readers in readers_list:
            os.environ["NWKNN_READERS"] = str(readers)
            for rep in range(2):
                acc.clear()
                shutil.rmtree(os.path.join(tmp, "partial")); os.makedirs(os.path.join(tmp, "partial"))
                with contextlib.redirect_stdout(open(os.devnull, "w")):
                    t0 = time.perf_counter()
                    cu_knn.compute_knn(tmp, "synthetic", d, "query.parquet", Q, "base.parquet", B, mem_tune=True, k=k, prescreen=pre)
                    wall = time.perf_counter() - t0
                if rep == 1 and not pre and readers == readers_list[0]:
                    pr = cProfile.Profile()
                    shutil.rmtree(os.path.join(tmp, "partial")); os.makedirs(os.path.join(tmp, "partial"))
                    with contextlib.redirect_stdout(open(os.devnull, "w")):
                        pr.enable()
                        cu_knn.compute_knn(tmp, "synthetic", d, "query.parquet", Q, "base.parquet", B, mem_tune=True, k=k, prescreen=pre)
                        pr.disable()
                    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
                print(f"B={B} prescreen={int(pre)} readers={readers} rep={rep}: wall {wall:.2f} s = {B * d * 4 / 1e9 / wall:.1f} GB/s of floats | " +
                      " | ".join(f"{n} {v:.2f}" for n, v in acc.items()), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
