"""``ck`` kNN stage -- the part of the reference's ``neighborhoodwatch/colbert_knn.py`` that is on the hot path.

    process_knn_computation(data_dir, base_filename, base_count, query_filename, query_count, mem_tune=False,
                            initial_batch_size=1000000, max_memory_threshold=0.1, k=100, split=False, engine='torch')

``ck`` treats every ColBERT token embedding (d = 128, columns ``token_embedding_<i>``, colbert_knn.py:229-233) as an
independent point and runs the same ``process_batches`` with ``engine='torch'`` (distance ``1 - q.b``), ``split=False`` and a
9000-row base tile when the requested tile exceeds 100 000 rows (colbert_knn.py:94-98,115-126): flat token-level kNN, not
MaxSim.  The ColBERT encoder, dataset download and the ``ck`` argument parser (colbert_knn.py:31-82,146-341) need model
checkpoints and the network and are out of scope.
"""
from __future__ import annotations

import math

import pyarrow.parquet as pq

from .cu_knn import process_batches, tune_memory


def process_knn_computation(data_dir,
                            base_filename,
                            base_count,
                            query_filename,
                            query_count,
                            mem_tune=False,
                            initial_batch_size=1000000,
                            max_memory_threshold=0.1,
                            k=100,
                            split=False,
                            engine='torch',
                            *,
                            device: int = 0,
                            prescreen: bool = False):
    if engine == 'torch' and initial_batch_size > 100000:
        initial_batch_size = 9000                                          # colbert_knn.py:95-98
    print("-- prepare query source table for brute-force KNN computation.")
    query_table = pq.read_table(query_filename).slice(0, query_count)       # colbert_knn.py:102-105: every column is used
    print(f"   query_table.shape: {query_table.shape}")
    base_table = pq.read_table(base_filename).slice(0, base_count)
    print(f"   base_table.shape: {base_table.shape}")
    batch_size = initial_batch_size
    if mem_tune:
        batch_size = tune_memory(base_table, batch_size, max_memory_threshold, None, device)
    batch_count = math.ceil(len(base_table) / batch_size)
    assert (len(base_table) % batch_size == 0) or k <= (len(base_table) % batch_size), \
        f"Cannot generate k of {k} with only {len(base_table)} rows and batch_size of {batch_size}."   # colbert_knn.py:112-113
    process_batches(data_dir, base_table, query_table, batch_count, batch_size, k, split, engine=engine, device=device,
                    prescreen=prescreen)
