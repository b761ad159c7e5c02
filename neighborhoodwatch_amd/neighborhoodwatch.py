"""``nw`` -- the reference's command line (neighborhoodwatch/neighborhoodwatch.py:30-199) over the HIP engine.

Positional arguments, every flag and the five-stage order are the reference's.  Two things differ by necessity:
  * stages 1-2 (embedding SQuAD / Wikipedia through remote model APIs, generate_dataset.py) need the network and
    paid services; here they write seeded synthetic embeddings with the same parquet schema and file names
    (``--synthetic-seed``), or are skipped when the source parquet files already exist (as the reference does,
    generate_dataset.py:266-268,298-300).  ``check_dataset_exists_remote`` (neighborhoodwatch.py:65) is not called.
  * stage 5 writes .fvec/.ivec with a vectorised writer (byte-identical to the reference's); hdf5 needs h5py and is
    skipped with a message where it is not installed.  ``--post-validation`` runs validate_files_v0 without the
    interactive prompt of neighborhoodwatch.py:184-186.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

from . import synth
from .cu_knn import compute_knn
from .cu_knn_ds import compute_knn_ds
from .merge import merge_indices_and_distances
from .parquet_to_format import generate_output_files, validate_files_v0
from .nw_utils import (get_effective_embedding_size, get_partial_distances_filename, get_partial_indices_filename,
                       get_source_base_dataset_filename, get_source_query_dataset_filename, get_valid_model_names_string,
                       get_model_prefix, is_valid_model_name, setup_model_output_folder)


def cleanup_partial_parquet(data_dir):
    """neighborhoodwatch.py:20-23."""
    for filename in os.listdir(data_dir):
        if filename.startswith(("distances", "indices", "final")):
            os.remove(f"{data_dir}/{filename}")


class KeepLineBreaksFormatter(argparse.RawTextHelpFormatter):
    pass


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(
        prog="nw",
        description="nw (neighborhood watch) uses GPU acceleration to generate ground truth KNN datasets",
        epilog="""
Some example commands:\n
    nw 1000 10000 -k 100 -m 'intfloat/e5-small-v2' --disable-memory-tuning
    nw 10000 1000000 -k 100 -m 'intfloat/e5-base-v2' --enable-memory-tuning
        """, formatter_class=KeepLineBreaksFormatter)
    parser.add_argument('query_count', type=int, help="number of query vectors to generate")
    parser.add_argument('base_count', type=int, help="number of base vectors to generate")
    parser.add_argument('-m', '--model_name', type=str,
                        help=f'model name (selects the embedding width) and must be one of: {get_valid_model_names_string()}')
    parser.add_argument('-ods', '--output_dimension_size', type=int, default=None,
                        help="Output dimension size; can be different from model's default dimension size!")
    parser.add_argument('-odt', '--output_dtype', type=str, default='float',
                        help="Output dtype; currently only valid for VoyageAI model!")
    parser.add_argument('-k', '--k', type=int, default=100, help='number of neighbors to compute per query vector')
    parser.add_argument('--data-dir', type=str, default='knn_dataset',
                        help='Directory to store the generated data (default: knn_dataset)')
    parser.add_argument('--use-dataset-api', action=argparse.BooleanOptionalAction, default=False,
                        help="Use 'pyarrow.dataset' API to read the dataset. Recommended for large datasets.")
    parser.add_argument('--gen-hdf5', action=argparse.BooleanOptionalAction, default=True,
                        help='Generate hdf5 files (default: True; needs h5py)')
    parser.add_argument('--post-validation', action=argparse.BooleanOptionalAction, default=False,
                        help='Validate the generated files (default: False)')
    parser.add_argument('--enable-memory-tuning', action='store_true', help='Enable memory tuning')
    parser.add_argument('--disable-memory-tuning', action='store_false',
                        help='Disable memory tuning (useful for very small datasets)')
    # additions (defaults keep the reference behaviour)
    parser.add_argument('--synthetic-seed', type=lambda s: int(s, 0), default=None,
                        help='seed of the synthetic embeddings written when the source parquet files do not exist')
    parser.add_argument('--normalize', action='store_true', help='synthetic rows are scaled to unit L2 norm')
    parser.add_argument('--embedding-layout', choices=['columns', 'list'], default='columns',
                        help="synthetic source files: 'columns' = embedding_0..d-1 (the reference's schema), "
                             "'list' = one FixedSizeList<float32, d> column 'embedding' (row-major; the kNN stage reads either)")
    parser.add_argument('--engine', choices=['raft', 'cuvs', 'torch'], default='raft',
                        help="distance of the reference engine of that name: sqeuclidean | cosine | 1-dot")
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--devices', type=lambda s: [int(x) for x in s.split(',')], default=None,
                        help='comma separated GPU ids: base tiles are dealt round-robin to them')
    parser.add_argument('--dtype', choices=['fp32', 'fp16'], default='fp32', help='compute path (fp32 is bit-exact; fp16 uses the fp16 MFMA kernel)')
    parser.add_argument('--prescreen', action='store_true',
                        help='fp32 results bit for bit, several times faster: fp16 MFMA pre-screen + canonical fp32 rescoring')
    parser.add_argument('--resume', action='store_true', help='keep existing partial files and skip their base tiles')
    return parser


def main(argv=None):
    start_time = time.time()
    args = build_parser().parse_args(argv)
    print(f"Neighborhood Watch (MI355X build): query count {args.query_count}, base count {args.base_count}, "
          f"model {args.model_name}, K {args.k}, dataset API {args.use_dataset_api}, memory tuning {args.enable_memory_tuning}")
    assert is_valid_model_name(args.model_name), \
        f"The given model name is invalid; must be one of: {get_valid_model_names_string()}"
    if args.model_name == 'colbertv2.0':
        raise SystemExit("For Colbert model, please use the `ck` program")                  # neighborhoodwatch.py:89-90

    data_dir = setup_model_output_folder(args.data_dir, args.model_name, args.query_count, args.base_count, args.k)
    output_dimension = get_effective_embedding_size(args.model_name, args.output_dimension_size)
    output_dtype = args.output_dtype if args.model_name.startswith('voyage') else None

    section = time.time()
    query_filename = get_source_query_dataset_filename(data_dir, args.model_name, args.query_count, output_dimension, output_dtype)
    if not os.path.exists(query_filename):
        seed = synth.SEED_QUERY if args.synthetic_seed is None else args.synthetic_seed + 1
        synth.write_source_parquet(query_filename, seed, args.query_count, output_dimension, args.normalize, layout=args.embedding_layout)
    print(f"(Query source file: {query_filename}) ({time.time() - section:.2f} s)")

    section = time.time()
    base_filename = get_source_base_dataset_filename(data_dir, args.model_name, args.base_count, output_dimension, output_dtype)
    if not os.path.exists(base_filename):
        seed = synth.SEED_BASE if args.synthetic_seed is None else args.synthetic_seed
        synth.write_source_parquet(base_filename, seed, args.base_count, output_dimension, args.normalize, layout=args.embedding_layout)
    print(f"(Base source file: {base_filename}) ({time.time() - section:.2f} s)")

    if not args.resume:
        cleanup_partial_parquet(f"{data_dir}/partial")                                  # neighborhoodwatch.py:130

    print("Computing knn ......")
    section = time.time()
    if args.use_dataset_api:
        compute_knn_ds(data_dir, output_dimension, query_filename, args.query_count, base_filename, args.base_count,
                       args.enable_memory_tuning, args.k, engine=args.engine, device=args.device, prescreen=args.prescreen,
                       resume=args.resume)
    else:
        compute_knn(data_dir, args.model_name, output_dimension, query_filename, args.query_count, base_filename,
                    args.base_count, args.enable_memory_tuning, args.k, engine=args.engine, device=args.device,
                    devices=args.devices, dtype=args.dtype, resume=args.resume, prescreen=args.prescreen)
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")

    print("Merging indices and distances ......")
    section = time.time()
    merge_indices_and_distances(data_dir, k=args.k, device=args.device)
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")
    print("Generating ivec's and fvec's ......")
    section = time.time()
    files = generate_output_files(data_dir, get_model_prefix(args.model_name), output_dimension, base_filename, query_filename,
                                  args.base_count, args.query_count, get_partial_indices_filename(data_dir, -1),
                                  get_partial_distances_filename(data_dir, -1), args.k, args.gen_hdf5, column_names=None,
                                  output_dtype=output_dtype)                               # neighborhoodwatch.py:167-179
    print(f"(Duration: {time.time() - section:.2f} seconds out of {time.time() - start_time:.2f} seconds)")
    if args.post_validation:
        print("Validating ivec's and fvec's ......")
        if validate_files_v0(data_dir, *files) != 0:          # the reference's own identity: 1 - q.b == distance / 2 on unit rows
            print("post-validation FAILED")
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
