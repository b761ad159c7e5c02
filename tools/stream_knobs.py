"""The streamed file path (compute_knn on the refcli shape from /dev/shm) by pyarrow pool x reader threads x gather threads, wall and CPU seconds: is the host split right?
usage: stream_knobs.py [base_rows]"""
import contextlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa, pyarrow.parquet as pq
from neighborhoodwatch_amd import KnnEngine, cu_knn
from neighborhoodwatch_amd.engine import default_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d, Q, k = 1536, 10_000, 100
tmp = tempfile.mkdtemp(prefix="nwk_knobs_", dir="/dev/shm")
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
    import pyarrow as pa
    for rep in range(3):
        for arrow_threads, readers, gather in ((16, 4, 8), (12, 4, 4), (8, 4, 4), (12, 3, 4), (12, 4, 8), (16, 4, 4)):
            os.environ["NWKNN_READERS"] = str(readers)
            pa.set_cpu_count(arrow_threads)
            default_engine(0).set_option("upload_threads", gather)
            shutil.rmtree(os.path.join(tmp, "partial"), ignore_errors=True); os.makedirs(os.path.join(tmp, "partial"))
            with contextlib.redirect_stdout(open(os.devnull, "w")):
                c0 = os.times(); t0 = time.perf_counter()
                cu_knn.compute_knn(tmp, "synthetic", d, "query.parquet", Q, "base.parquet", B, mem_tune=True, k=k)
                wall = time.perf_counter() - t0; c1 = os.times()
            print(f"rep {rep} arrow pool {arrow_threads:2d} readers {readers} gather threads {gather:2d}: {wall:.2f} s wall, {c1.user + c1.system - c0.user - c0.system:5.1f} CPU-s", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
