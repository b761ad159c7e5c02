"""(no GPU) What the host really gives this process (the GPU box shows 256 CPUs and a cgroup quota of 16), and what parquet decode costs in
CPU-SECONDS per GB -- the resource the file path is bound by -- under pyarrow's memory pools and CPU pool sizes.
usage: arrow_pool_probe.py [rows] [dir]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, pyarrow as pa, pyarrow.parquet as pq

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
root = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
d = 1536
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, "=", open(f).read().strip()[:200])
    except OSError as e:
        print(f, "-", e.strerror)
print(f"os.cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, arrow cpu pool {pa.cpu_count()}, io pool {pa.io_thread_count()}, "
      f"default memory pool {pa.default_memory_pool().backend_name}, backends {pa.supported_memory_backends()}", flush=True)

names = [f"embedding_{j}" for j in range(d)]
rng = np.random.default_rng(1)
path = os.path.join(root, "nwk_pool_probe.parquet")
w = None
for r0 in range(0, rows, 100_000):
    n = min(100_000, rows - r0)
    cols = rng.random((d, n), dtype=np.float32) * 2 - 1
    t = pa.Table.from_arrays([pa.array(cols[j]) for j in range(d)], names=names)
    if w is None:
        w = pq.ParquetWriter(path, t.schema, compression="NONE", use_dictionary=False, write_statistics=False)
    w.write_table(t)
w.close()
del cols, t
raw_gb = rows * d * 4 / 1e9
n_groups = pq.read_metadata(path).num_row_groups


def cpu_s():
    t = os.times()
    return t.user + t.system, t.system


def decode(n_readers, hold):
    """n_readers threads, each decoding its share of the row groups; `hold` = keep the last decoded table alive while the next is decoded (the
    file path holds a few pieces: decoded, being gathered, being released)."""
    def work(t):
        f = pq.ParquetFile(path, memory_map=True)
        keep = None
        for g in range(t, n_groups, n_readers):
            nxt = f.read_row_groups([g], use_threads=True)
            if hold:
                keep = nxt
            del nxt
        del keep
    c0, s0 = cpu_s(); t0 = time.perf_counter()
    with ThreadPoolExecutor(n_readers) as pool:
        list(pool.map(work, range(n_readers)))
    dt = time.perf_counter() - t0
    c1, s1 = cpu_s()
    return dt, c1 - c0, s1 - s0


pools = [("default", None)]
if "jemalloc" in pa.supported_memory_backends():
    pools.append(("jemalloc", pa.jemalloc_memory_pool))
if "mimalloc" in pa.supported_memory_backends():
    pools.append(("mimalloc", pa.mimalloc_memory_pool))
pools.append(("system", pa.system_memory_pool))
try:
    for pool_name, make in pools:
        if make is not None:
            pa.set_memory_pool(make())
        for cpu_pool in (256, 16):
            pa.set_cpu_count(cpu_pool)
            for n_readers in (3, 4):
                res = [decode(n_readers, True) for _ in range(4)]
                best = min(r[0] for r in res)
                print(f"memory pool {pool_name:8s} arrow cpu pool {cpu_pool:3d}, {n_readers} readers: wall " + " ".join(f"{r[0]:.2f}" for r in res) +
                      f" s (best {raw_gb / best:5.1f} GB/s) | CPU-s per GB " + " ".join(f"{r[1] / raw_gb:.2f}" for r in res) +
                      " | of which kernel " + " ".join(f"{r[2] / raw_gb:.2f}" for r in res), flush=True)
finally:
    os.remove(path)
