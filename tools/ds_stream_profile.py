"""cProfile of compute_knn_ds on the refcli shape (10k x B x 1536 from /dev/shm): where the dataset flavour spends what compute_knn does not.
usage: ds_stream_profile.py [base_rows]"""
import contextlib, cProfile, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa, pyarrow.parquet as pq
from neighborhoodwatch_amd import KnnEngine, cu_knn, cu_knn_ds

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, Q, k = 1536, 10_000, 100
tmp = tempfile.mkdtemp(prefix="nwk_ds_", dir="/dev/shm")
try:
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
    for rep in range(3):
        for how in ("compute_knn", "compute_knn_ds"):
            shutil.rmtree(os.path.join(tmp, "partial"), ignore_errors=True); os.makedirs(os.path.join(tmp, "partial"))
            pr = cProfile.Profile() if rep == 2 else None
            with contextlib.redirect_stdout(open(os.devnull, "w")):
                t0 = time.perf_counter()
                if pr: pr.enable()
                if how == "compute_knn":
                    cu_knn.compute_knn(tmp, "synthetic", d, "query.parquet", Q, "base.parquet", B, mem_tune=True, k=k)
                else:
                    cu_knn_ds.compute_knn_ds(tmp, d, "query.parquet", Q, "base.parquet", B, mem_tune=True, k=k)
                if pr: pr.disable()
                wall = time.perf_counter() - t0
            print(f"{how} rep {rep}: {wall:.2f} s", flush=True)
            if pr:
                pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
