#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Run from the repo root in the build container:

    python tests/golden/make_golden.py

Two kinds of fixtures (data only -- inputs by seed, expected outputs as arrays):

1. ``merge_*.npz`` -- produced by RUNNING THE REFERENCE's own ``neighborhoodwatch/merge.py``
   (``merge_indices_and_distances``, merge.py:39-114) from /root/reference on seeded partial lists.
   merge.py is pure numpy/pandas/pyarrow; its module-level ``import cudf`` (merge.py:2, never used)
   and the embedding-vendor imports pulled in through nw_utils -> model_generator are satisfied with
   inert ``MagicMock`` modules, so every byte of the output comes from the reference's code.  The
   reference itself is not copied anywhere and never travels to the GPU box.

2. ``knn_*.npz`` -- the kNN operator lives in CUDA-only wheels (pylibraft/cuvs/torch-CUDA) that cannot
   run here, and the reference's tests hold no expected values, so these vectors pin the ORACLE
   (oracle/nwknn_oracle.c) instead: canonical-arithmetic outputs, the float64 exact ranking, and --
   for the ``engine='torch'`` formula -- a literal torch-CPU evaluation of cu_knn.py:256-257
   (``1 - torch.matmul(q, b.T)``; ``torch.topk(.., k, dim=1, largest=False)``).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import sys
import tempfile
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import cpu_oracle as orc  # noqa: E402

OUT = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")


# ------------------------------------------------------------------------------------------------
# merge fixtures from the real reference
# ------------------------------------------------------------------------------------------------
def _import_reference_merge():
    sys.dont_write_bytecode = True
    for m in ["cudf", "openai", "sentence_transformers", "vertexai", "vertexai.preview",
              "vertexai.preview.language_models", "vertexai.language_models", "colbert", "colbert.infra",
              "colbert.infra.config", "colbert.modeling", "colbert.modeling.checkpoint", "colbert.indexing",
              "colbert.indexing.collection_encoder", "cohere", "voyageai", "spacy"]:
        sys.modules.setdefault(m, MagicMock())
    sys.path.insert(0, str(REFERENCE))
    import neighborhoodwatch.merge as ref_merge  # type: ignore
    return ref_merge


def _sorted_partial(rng, q, k, lo, n_rows, tie_pool=None):
    """One partial pair shaped like cu_knn.py:271-285 output: ascending float32 distances, int64 ids."""
    if tie_pool is None:
        d = np.sort(rng.random((q, k)).astype(np.float32) * 4.0, axis=1)
    else:
        d = np.sort(rng.choice(tie_pool, size=(q, k)).astype(np.float32), axis=1)
    i = np.stack([rng.choice(n_rows, size=k, replace=False) for _ in range(q)]).astype(np.int64) + lo
    return d, i


def make_merge_case(ref_merge, name, seed, q, k, n_lists, ties=False, rownum=False):
    rng = np.random.default_rng(seed)
    pool = (np.arange(1, 13, dtype=np.float32) * np.float32(0.25)) if ties else None
    ds, is_ = [], []
    for l in range(n_lists):
        d, i = _sorted_partial(rng, q, k, l * 1000, 1000, pool)
        ds.append(d); is_.append(i)
    tmp = Path(tempfile.mkdtemp(prefix="nwgold_"))
    try:
        (tmp / "partial").mkdir()
        for l in range(n_lists):
            cols = [str(c) for c in range(k)]
            ddf = pd.DataFrame(ds[l], columns=cols)
            idf = pd.DataFrame(is_[l], columns=cols)
            if rownum:   # tests/test_03_merge.py:14-24 adds a RowNum column; merge.py:33-34 "removes" it (a no-op)
                pass
            ddf.to_parquet(tmp / "partial" / f"distances{l}.parquet")
            idf.to_parquet(tmp / "partial" / f"indices{l}.parquet")
        ref_merge.merge_indices_and_distances(str(tmp), k=k)
        fd = pd.read_parquet(tmp / "partial" / "final_distances.parquet")
        fi = pd.read_parquet(tmp / "partial" / "final_indices.parquet")
        np.savez_compressed(OUT / f"merge_{name}.npz",
                            in_dist=np.stack(ds), in_idx=np.stack(is_),
                            out_dist=fd.values.astype(np.float32), out_idx=fi.values.astype(np.int64),
                            out_columns=np.array(list(fd.columns)), out_dist_dtype=str(fd.dtypes.iloc[0]),
                            out_idx_dtype=str(fi.dtypes.iloc[0]), k=k, seed=seed)
        print(f"merge_{name}: Q={q} k={k} lists={n_lists} ties={ties} -> {fd.shape} {fd.dtypes.iloc[0]} {fi.dtypes.iloc[0]}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------
# kNN fixtures (oracle pins)
# ------------------------------------------------------------------------------------------------
def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_knn_case(name, base, queries, k, metric, keep_rows=None, torch_literal=False, note=""):
    d32, i32 = orc.knn_canonical(base, queries, k, metric)
    d32s, i32s = orc.knn_canonical(base, queries, k, metric, scalar=True)
    assert np.array_equal(d32, d32s) and np.array_equal(i32, i32s), "AVX2 and scalar oracle paths disagree"
    d64, i64 = orc.knn_exact_fp64(base, queries, k, metric)
    frac = float((i32 == i64).mean())
    err, slack = orc.check_against_exact(base, queries, d32, i32, d64, metric)
    extra = {}
    if torch_literal:
        import torch
        qt, bt = torch.from_numpy(np.ascontiguousarray(queries)), torch.from_numpy(np.ascontiguousarray(base))
        distance_batch = 1 - torch.matmul(qt, bt.T)                       # cu_knn.py:256
        dt, it = torch.topk(distance_batch, k, dim=1, largest=False)      # cu_knn.py:257
        extra = dict(torch_dist=dt.numpy(), torch_idx=it.numpy().astype(np.int64))
    rows = slice(None) if keep_rows is None else slice(0, keep_rows)
    np.savez_compressed(OUT / f"knn_{name}.npz", k=k, metric=metric, note=note,
                        canon_dist=d32[rows], canon_idx=i32[rows].astype(np.int32),
                        exact_dist=d64[rows], exact_idx=i64[rows].astype(np.int32),
                        canon_dist_sha256=_sha(d32), canon_idx_sha256=_sha(i32.astype(np.int64)),
                        **extra)
    print(f"knn_{name}: {queries.shape[0]}x{base.shape[0]} d={base.shape[1]} k={k} {metric}: "
          f"idx==fp64 {frac:.5f} (all mismatches are fp32 near-ties), max|d-d64|={err:.3e}, rank slack {slack:.2e}")


def make_ifvec_case():
    """fvec/ivec bytes written by the reference's own write_ivec_fvec_from_dataframe (parquet_to_format.py:71-89).
    parquet_to_format.py imports cupy / h5py / cuvs at module level for its hdf5 writer and GPU validators only; the
    writer itself is pure pandas/numpy, so those three are satisfied with inert mocks."""
    for m in ["cupy", "h5py", "cuvs", "cuvs.distance", "cuvs.neighbors"]:
        sys.modules.setdefault(m, MagicMock())
    import neighborhoodwatch.parquet_to_format as ref_ptf  # type: ignore
    rng = np.random.default_rng(21)
    vec = rng.standard_normal((23, 7)).astype(np.float32)
    ids = rng.integers(0, 2 ** 31 - 1, size=(23, 5)).astype(np.int64)
    tmp = Path(tempfile.mkdtemp(prefix="nwgold_"))
    try:
        ref_ptf.write_ivec_fvec_from_dataframe(str(tmp), "m", "v.fvec", pd.DataFrame(vec), "f", 7)
        ref_ptf.write_ivec_fvec_from_dataframe(str(tmp), "m", "i.ivec", pd.DataFrame(ids), "i", 5)
        idr = pd.DataFrame(ids); idr["RowNum"] = np.arange(23)                     # trailing RowNum column is dropped (:77-78)
        ref_ptf.write_ivec_fvec_from_dataframe(str(tmp), "m", "r.ivec", idr, "i", 5)
        np.savez_compressed(OUT / "ifvec_reference_writer.npz", vec=vec, ids=ids,
                            fvec_bytes=np.frombuffer((tmp / "v.fvec").read_bytes(), dtype=np.uint8),
                            ivec_bytes=np.frombuffer((tmp / "i.ivec").read_bytes(), dtype=np.uint8),
                            ivec_rownum_bytes=np.frombuffer((tmp / "r.ivec").read_bytes(), dtype=np.uint8),
                            count=ref_ptf.count_vectors(str(tmp), "v.fvec"), third=np.array(ref_ptf.get_nth_vector(str(tmp), "v.fvec", 3)),
                            third_ids=np.array(ref_ptf.get_nth_vector(str(tmp), "i.ivec", 3)))
        print(f"ifvec_reference_writer: fvec {(tmp / 'v.fvec').stat().st_size} B, ivec {(tmp / 'i.ivec').stat().st_size} B")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ref_merge = _import_reference_merge()
    make_ifvec_case()
    make_merge_case(ref_merge, "basic", 11, q=50, k=10, n_lists=3)
    make_merge_case(ref_merge, "ties", 12, q=40, k=8, n_lists=4, ties=True)
    make_merge_case(ref_merge, "single", 13, q=17, k=5, n_lists=1)
    make_merge_case(ref_merge, "nine", 14, q=16, k=100, n_lists=9)     # docs/cli.png shape: 9 partial pairs, k=100

    # G1: the shape of the reference's tests/test_02_cu_knn.py:68-71 (Q=100, B=1000, d=384, k=10, unit rows)
    b = orc.gen_unit_rows(0x6E770011, 0, 1000, 384); q = orc.gen_unit_rows(0x6E770012, 0, 100, 384)
    make_knn_case("g1_unit_sqeuclidean", b, q, 10, "sqeuclidean", note="gen_unit_rows seeds 0x6E770011/12")
    make_knn_case("g1_unit_one_minus_dot", b, q, 10, "one_minus_dot", torch_literal=True)
    make_knn_case("g1_unit_cosine", b, q, 10, "cosine")
    # G2: BASELINE config 1 (1k x 10k, d=384, k=100), generator values in [-1,1)
    b = orc.gen(orc.SEED_BASE, 0, 10000, 384); q = orc.gen(orc.SEED_QUERY, 0, 1000, 384)
    make_knn_case("g2_c1_sqeuclidean", b, q, 100, "sqeuclidean", keep_rows=32, note="gen SEED_BASE/SEED_QUERY")
    # G3: adversarial -- exact duplicates, d not a multiple of the MFMA K tile, k=1 and k=B
    b = orc.gen(0x6E770031, 0, 700, 100); b[350:700] = b[0:350]; q = np.concatenate([orc.gen(0x6E770032, 0, 45, 100), b[:5]])
    make_knn_case("g3_dups_d100", b, q, 16, "sqeuclidean", note="rows 350..699 duplicate rows 0..349; last 5 queries are base rows")
    make_knn_case("g3_k1", b, q, 1, "sqeuclidean")
    b = orc.gen(0x6E770033, 0, 130, 36); q = orc.gen(0x6E770034, 0, 9, 36)
    make_knn_case("g3_k_equals_b", b, q, 130, "sqeuclidean")


if __name__ == "__main__":
    main()
