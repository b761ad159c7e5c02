"""How fast can the kNN stage's source parquet be decoded, and with which reader arrangement?  (no GPU needed)
One float32 column per dimension (the reference's layout, generate_dataset.py:219-262), 100 000-row row groups, written twice: uncompressed
/ no dictionary (bench.py's first file leg) and with pyarrow's defaults -- snappy + dictionary -- as the reference's ParquetStreamer writes
(generate_dataset.py:241-242).  Timed: row groups decoded one after the other with pyarrow's column threads, and N reader threads with a
ParquetFile handle each (row group g to thread g mod N), with and without column threads inside each.
usage: parquet_decode_probe.py [rows] [d] [dir]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
root = sys.argv[3] if len(sys.argv) > 3 else ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
print(f"cpus {os.cpu_count()}, arrow cpu pool {pa.cpu_count()}, io pool {pa.io_thread_count()}, rows {rows}, d {d}, dir {root}", flush=True)
names = [f"embedding_{j}" for j in range(d)]
rng = np.random.default_rng(1)


def write(path, **kw):
    w = None
    for r0 in range(0, rows, 100_000):
        n = min(100_000, rows - r0)
        cols = rng.random((d, n), dtype=np.float32) * 2 - 1
        t = pa.Table.from_arrays([pa.array(cols[j]) for j in range(d)], names=names)
        if w is None:
            w = pq.ParquetWriter(path, t.schema, **kw)
        w.write_table(t)
    w.close()


files = {"uncompressed": dict(compression="NONE", use_dictionary=False, write_statistics=False), "pyarrow defaults (snappy, dictionary)": {}}
for label, kw in files.items():
    path = os.path.join(root, "nwk_decode_probe.parquet")
    t0 = time.perf_counter()
    write(path, **kw)
    gb = os.path.getsize(path) / 1e9
    raw_gb = rows * d * 4 / 1e9
    print(f"== {label}: file {gb:.2f} GB for {raw_gb:.2f} GB of floats (written in {time.perf_counter() - t0:.1f} s)", flush=True)
    n_groups = pq.read_metadata(path).num_row_groups

    def timed(what, fn):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print(f"   {what:72s} {dt:6.2f} s   {raw_gb / dt:5.2f} GB/s of floats", flush=True)

    def sequential(mm):
        f = pq.ParquetFile(path, memory_map=mm)
        for g in range(n_groups):
            f.read_row_groups([g], use_threads=True)
    timed("one reader, row group after row group, column threads", lambda: sequential(False))
    timed("... memory-mapped", lambda: sequential(True))
    timed("pq.read_table (whole file, column threads)", lambda: pq.read_table(path))
    for n_readers in (2, 4, 8, 16):
        if n_readers > max(2, os.cpu_count()):
            break
        for inner in (True, False):
            def parallel():
                def work(t):
                    f = pq.ParquetFile(path, memory_map=True)
                    for g in range(t, n_groups, n_readers):
                        f.read_row_groups([g], use_threads=inner)
                with ThreadPoolExecutor(n_readers) as pool:
                    list(pool.map(work, range(n_readers)))
            timed(f"{n_readers} readers (own handle, memory-mapped), column threads {'on' if inner else 'off'}", parallel)
    os.remove(path)
