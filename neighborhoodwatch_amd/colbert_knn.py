"""``ck`` kNN stage -- the part of the reference's ``neighborhoodwatch/colbert_knn.py`` that is on the hot path.

    process_knn_computation(data_dir, base_filename, base_count, query_filename, query_count, mem_tune=False,
                            initial_batch_size=1000000, max_memory_threshold=0.1, k=100, split=False, engine='torch')

``ck`` treats every ColBERT token embedding (d = 128, columns ``token_embedding_<i>``, colbert_knn.py:229-233) as an
independent point and runs the same ``process_batches`` with ``engine='torch'`` (distance ``1 - q.b``), ``split=False`` and a
9000-row base tile when the requested tile exceeds 100 000 rows (colbert_knn.py:94-98,115-126): flat token-level kNN, not
MaxSim.  The ColBERT encoder and the dataset download (colbert_knn.py:31-82, model_generator.py:392-439) need model
checkpoints and the network and are out of scope; ``main`` keeps the ``ck`` argument parser and stage order
(colbert_knn.py:146-341) and writes seeded synthetic unit-norm token rows where the reference runs the encoder.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import pyarrow.parquet as pq

from . import synth
from .cu_knn import process_batches, tune_memory
from .merge import merge_indices_and_distances
from .nw_utils import (get_effective_embedding_size, get_full_filename, get_model_prefix, get_partial_distances_filename,
                       get_partial_indices_filename, setup_model_output_folder)
from .parquet_to_format import generate_output_files

COLBERT_MODEL = "colbertv2.0"                    # model_generator.py:26-58 (EmbeddingModelName.COLBERT_V2)


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


def build_parser() -> argparse.ArgumentParser:
    """The reference's ``ck`` flags (colbert_knn.py:149-172), plus the offline / engine additions of ``nw``."""
    parser = argparse.ArgumentParser(
        description="ck (Colbert KNN) generates ground truth KNN datasets over Colbert token embeddings (MI355X build).",
        epilog="Example:\n    ck 100000 1000000 -k 100 --disable-memory-tuning", formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('query_token_count', type=int, help="number of query token vectors to generate")
    parser.add_argument('base_token_count', type=int, help="number of base token vectors to generate")
    parser.add_argument('-m', '--model_name', type=str, default=COLBERT_MODEL, help='Colbert model name (default: colbertv2.0)')
    parser.add_argument('-k', '--k', type=int, default=100, help='number of neighbors to compute per query vector')
    parser.add_argument('-es', '--embedding-scale', type=str, default="medium",
                        help='Embedding scale. Options: small (10000), medium(100000), large (1000000) (default: medium)')
    parser.add_argument('--data-dir', type=str, default='knn_dataset', help='Directory to store the generated data (default: knn_dataset)')
    parser.add_argument('--use-dataset-api', action=argparse.BooleanOptionalAction, default=False,
                        help="Use 'pyarrow.dataset' API to read the dataset. Recommended for large datasets.")
    parser.add_argument('--gen-hdf5', action=argparse.BooleanOptionalAction, default=True, help='Generate hdf5 files (default: True; needs h5py)')
    parser.add_argument('--enable-memory-tuning', action='store_true', help='Enable memory tuning')
    parser.add_argument('--disable-memory-tuning', action='store_false', help='Disable memory tuning (useful for very small datasets)')
    parser.add_argument('--engine', type=str, default='torch', help='Pick raft, cuvs, or torch as the brute force knn engine')
    # additions (defaults keep the reference behaviour)
    parser.add_argument('--synthetic-seed', type=lambda s: int(s, 0), default=None,
                        help='seed of the synthetic token embeddings written when the source parquet files do not exist')
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--prescreen', action='store_true',
                        help='the same results bit for bit, several times faster: fp16 MFMA pre-screen + canonical fp32 rescoring')
    return parser


def main(argv=None):
    start_time = time.time()
    args = build_parser().parse_args(argv)
    if args.embedding_scale not in (None, "small", "medium", "large"):
        print(f"Invalid embedding scale: {args.embedding_scale}")                   # colbert_knn.py:202-210
        return 1
    assert args.model_name == COLBERT_MODEL, "`ck` program is reserved for Colbert model..."        # colbert_knn.py:196-197
    print(f"Colbert KNN (MI355X build): query token vector count {args.query_token_count}, base token vector count "
          f"{args.base_token_count}, model {args.model_name}, K {args.k}, engine {args.engine}, memory tuning {args.enable_memory_tuning}")
    model_prefix = get_model_prefix(args.model_name)
    data_dir = setup_model_output_folder(args.data_dir, args.model_name, args.query_token_count, args.base_token_count, args.k)
    input_dimensions = get_effective_embedding_size(args.model_name)
    token_embed_columns = [f'token_embedding_{i}' for i in range(input_dimensions)]
    seed = synth.SEED_BASE if args.synthetic_seed is None else args.synthetic_seed

    section = time.time()
    query_file = get_full_filename(data_dir, f"{model_prefix}_{input_dimensions}_query_token{args.query_token_count}_src.parquet")
    if not os.path.exists(query_file):                                             # colbert_knn.py:235
        synth.write_source_parquet(query_file, seed + 1, args.query_token_count, input_dimensions, normalize=True,
                                   column_prefix="token_embedding_", metadata_columns=False)
    else:
        print("The source query embed file already exists, skip processing the query source dataset.")
    print(f"(Query token file: {query_file}) ({time.time() - section:.2f} s)")
    section = time.time()
    base_file = get_full_filename(data_dir, f"{model_prefix}_{input_dimensions}_base_token{args.base_token_count}_src.parquet")
    if not os.path.exists(base_file):                                              # colbert_knn.py:266
        synth.write_source_parquet(base_file, seed, args.base_token_count, input_dimensions, normalize=True,
                                   column_prefix="token_embedding_", metadata_columns=False)
    else:
        print("The source base embed file already exists, skip processing the base source dataset.")
    print(f"(Base token file: {base_file}) ({time.time() - section:.2f} s)")

    for stale in os.listdir(f"{data_dir}/partial") if os.path.isdir(f"{data_dir}/partial") else []:
        if stale.endswith(".parquet"):                                             # as `nw` does (neighborhoodwatch.py:20-23,130); the
            os.remove(f"{data_dir}/partial/{stale}")                               # reference's `ck` merges whatever an older run left behind
    print("Computing knn ......")
    section = time.time()
    process_knn_computation(data_dir, base_file, args.base_token_count, query_file, args.query_token_count,
                            mem_tune=args.enable_memory_tuning, k=args.k, engine=args.engine, device=args.device,
                            prescreen=args.prescreen)                              # colbert_knn.py:308-315
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")
    print("Merging indices and distances ......")
    section = time.time()
    merge_indices_and_distances(data_dir, k=args.k, device=args.device)           # colbert_knn.py:321
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")
    print("Generating ivec's and fvec's ......")
    section = time.time()
    generate_output_files(data_dir, model_prefix, input_dimensions, base_file, query_file, args.base_token_count,
                          args.query_token_count, get_partial_indices_filename(data_dir, -1), get_partial_distances_filename(data_dir, -1),
                          args.k, args.gen_hdf5, token_embed_columns)               # colbert_knn.py:328-339
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
