"""Where the file API's kNN stage spends its time (C2 size): parquet read, column upload + prep, search, partial-file write.
usage: file_path_times.py <data_dir with the nw layout> <query parquet> <base parquet> [prescreen 0|1]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pyarrow.parquet as pq
from neighborhoodwatch_amd import KnnEngine
from neighborhoodwatch_amd.cu_knn import stream_cudf_to_parquet, table_columns
qf, bf = sys.argv[2], sys.argv[3]
pre = len(sys.argv) > 4 and sys.argv[4] == "1"
t = time.perf_counter(); qt = pq.read_table(qf); bt = pq.read_table(bf); t_read = time.perf_counter() - t
qt = qt.select([c for c in qt.schema.names if c.startswith("embedding_")]); bt = bt.select([c for c in bt.schema.names if c.startswith("embedding_")])
eng = KnnEngine(0); eng.set_prescreen(pre)
for rep in range(2):
    t = time.perf_counter(); eng.set_base_columns(table_columns(bt)); t_up = time.perf_counter() - t
    t = time.perf_counter(); d, i = eng.search_columns(table_columns(qt), 100); t_s = time.perf_counter() - t
t = time.perf_counter(); stream_cudf_to_parquet("/tmp/_d.parquet", d, 100000); stream_cudf_to_parquet("/tmp/_i.parquet", i, 100000); t_w = time.perf_counter() - t
print(f"read parquet {t_read:.3f} s | upload+prep base ({bt.num_rows} x {bt.num_columns}) {t_up:.3f} s | upload queries + search {t_s:.3f} s | write partial pair {t_w:.3f} s")
