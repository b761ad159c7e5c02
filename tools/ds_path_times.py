"""The file paths at C2 size (10k queries x 1M base rows, d = 768, float32 columns, 100k-row row groups): wall time of the kNN stage with
the reader one tile ahead of the GPU (PREFETCH_SLOTS = 2, the product) against strictly serial read -> search (PREFETCH_SLOTS = 1, how the
reference and round 2 ran), next to the two things being overlapped: decoding the base file and searching it.
usage: ds_path_times.py [base_rows] [batch_rows] [prescreen 0|1]"""
import os, resource, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pyarrow.parquet as pq
from neighborhoodwatch_amd import KnnEngine, cu_knn, cu_knn_ds, merge, nw_utils, synth
from neighborhoodwatch_amd.parquet_stream import ParquetRows

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
pre = len(sys.argv) > 3 and sys.argv[3] == "1"
Q, d, k = 10_000, 768, 100
root = tempfile.mkdtemp(prefix="nwds_", dir="/tmp")
data_dir = nw_utils.setup_model_output_folder(root, "intfloat/e5-base-v2", Q, B, k)
qf = nw_utils.get_source_query_dataset_filename(data_dir, "intfloat/e5-base-v2", Q, d)
bf = nw_utils.get_source_base_dataset_filename(data_dir, "intfloat/e5-base-v2", B, d)
t = time.perf_counter()
synth.write_source_parquet(qf, synth.SEED_QUERY, Q, d); synth.write_source_parquet(bf, synth.SEED_BASE, B, d, chunk_rows=100_000)
print(f"wrote {os.path.getsize(bf) / 1e9:.2f} GB base file in {time.perf_counter() - t:.1f} s", flush=True)
KnnEngine(0).close()                                               # GPU context up before anything is timed

rows = ParquetRows(bf, B, [f"embedding_{i}" for i in range(d)])
t = time.perf_counter()
for tb in rows.iter_tables(batch):
    cu_knn.table_columns(tb)
t_read = time.perf_counter() - t
print(f"decode the base file alone, {batch}-row batches: {t_read:.2f} s", flush=True)


def run(fn, slots):
    cu_knn.PREFETCH_SLOTS = slots; cu_knn_ds.PREFETCH_SLOTS = slots
    for f in os.listdir(f"{data_dir}/partial"):
        os.remove(f"{data_dir}/partial/{f}")
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


ds = lambda: cu_knn_ds.compute_knn_ds(data_dir, d, os.path.basename(qf), Q, os.path.basename(bf), B, False, k, batch, prescreen=pre)
tb = lambda: cu_knn.compute_knn(data_dir, "intfloat/e5-base-v2", d, os.path.basename(qf), Q, os.path.basename(bf), B, False, k, batch, prescreen=pre)
run(ds, 2)                                                         # warm-up: page cache, pinned buffers, kernels
for name, fn in (("compute_knn_ds", ds), ("compute_knn (+ partial pairs)", tb)):
    for rnd in range(2):
        serial, overlapped = run(fn, 1), run(fn, 2)
        print(f"{name}: serial read -> search {serial:.2f} s | reader one batch ahead {overlapped:.2f} s   (prescreen={int(pre)}, round {rnd})", flush=True)
import pyarrow as pa
print(f"arrow memory pool high-water mark over the whole script: {pa.default_memory_pool().max_memory() / 1e9:.2f} GB with {batch * d * 4 / 1e9:.2f} GB per decoded batch "
      f"(base file decoded: {B * d * 4 / 1e9:.2f} GB; the process's peak RSS, {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB, is set by writing the source files)")
