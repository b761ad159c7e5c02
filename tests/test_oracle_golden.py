"""CPU: pin the oracle.  kNN vectors pin oracle/nwknn_oracle.c (canonical arithmetic, fp64 truth, literal torch
formula of cu_knn.py:256-257); merge vectors were produced by the reference's own merge.py (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN

KNN_CASES = {
    "g1_unit_sqeuclidean": lambda o: (o.gen_unit_rows(0x6E770011, 0, 1000, 384), o.gen_unit_rows(0x6E770012, 0, 100, 384)),
    "g1_unit_one_minus_dot": lambda o: (o.gen_unit_rows(0x6E770011, 0, 1000, 384), o.gen_unit_rows(0x6E770012, 0, 100, 384)),
    "g1_unit_cosine": lambda o: (o.gen_unit_rows(0x6E770011, 0, 1000, 384), o.gen_unit_rows(0x6E770012, 0, 100, 384)),
    "g2_c1_sqeuclidean": lambda o: (o.gen(o.SEED_BASE, 0, 10000, 384), o.gen(o.SEED_QUERY, 0, 1000, 384)),
    "g3_dups_d100": None, "g3_k1": None, "g3_k_equals_b": None,
}


def golden_inputs(o, name):
    """Inputs are regenerated from seeds (the fixtures hold outputs only)."""
    if name in ("g3_dups_d100", "g3_k1"):
        b = o.gen(0x6E770031, 0, 700, 100); b[350:700] = b[0:350]
        q = np.concatenate([o.gen(0x6E770032, 0, 45, 100), b[:5]])
        return b, q
    if name == "g3_k_equals_b":
        return o.gen(0x6E770033, 0, 130, 36), o.gen(0x6E770034, 0, 9, 36)
    return KNN_CASES[name](o)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(KNN_CASES))
def test_oracle_reproduces_golden_knn(oracle, name):
    g = np.load(GOLDEN / f"knn_{name}.npz")
    base, queries = golden_inputs(oracle, name)
    k, metric = int(g["k"]), str(g["metric"])
    d, i = oracle.knn_canonical(base, queries, k, metric)
    n = g["canon_idx"].shape[0]
    assert np.array_equal(i[:n], g["canon_idx"].astype(np.int64))
    assert np.array_equal(d[:n].view(np.uint32), g["canon_dist"].view(np.uint32))
    assert _sha(d) == str(g["canon_dist_sha256"]) and _sha(i) == str(g["canon_idx_sha256"])
    # against the float64 truth: distances within 1e-5 (relative to |q|^2+|b|^2), ranks equal modulo fp32 near-ties
    exact_d, _ = oracle.knn_exact_fp64(base, queries, k, metric)
    assert np.allclose(exact_d[:n], g["exact_dist"], rtol=1e-12, atol=1e-12)
    oracle.check_against_exact(base, queries, d, i, exact_d, metric)


def test_scalar_and_simd_oracle_paths_agree(oracle):
    base, queries = oracle.gen(7, 0, 999, 77), oracle.gen(8, 0, 33, 77)
    for metric in oracle.METRICS:
        d1, i1 = oracle.knn_canonical(base, queries, 20, metric)
        d2, i2 = oracle.knn_canonical(base, queries, 20, metric, scalar=True)
        assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))


def test_canonical_dot_is_the_ascending_fmaf_chain(oracle):
    """The contract the MFMA K-loop fulfils: acc = fma(q_k, b_k, acc), k ascending, one rounding per step."""
    rng = np.random.default_rng(3)
    q = rng.standard_normal(97).astype(np.float32); b = rng.standard_normal(97).astype(np.float32)
    acc = np.float32(0)
    for a, c in zip(q, b):          # exact product + one rounding == fma, done in float64 (24x24 bits fit; sum of two fits 53 when close)
        acc = np.float32(np.float64(a) * np.float64(c) + np.float64(acc))
    d, _ = oracle.knn_canonical(b[None, :], q[None, :], 1, "one_minus_dot")
    assert d[0, 0] == np.float32(1.0) - acc


def test_torch_literal_formula_agrees_within_tolerance(oracle):
    """cu_knn.py:256-257 evaluated literally by torch on CPU (stored in the fixture) vs the canonical oracle."""
    g = np.load(GOLDEN / "knn_g1_unit_one_minus_dot.npz")
    assert np.abs(g["torch_dist"] - g["canon_dist"]).max() <= 1e-5
    assert (g["torch_idx"] == g["canon_idx"]).mean() > 0.99
    base, queries = golden_inputs(oracle, "g1_unit_one_minus_dot")
    exact_d, _ = oracle.knn_exact_fp64(base, queries, int(g["k"]), "one_minus_dot")
    oracle.check_against_exact(base, queries, g["torch_dist"], g["torch_idx"], exact_d, "one_minus_dot")


def test_sqeuclidean_is_twice_one_minus_dot_on_unit_rows(oracle):
    """The only value-level check in the reference: validate_files_v0, parquet_to_format.py:368-373,
    isclose(1 - dot(q, b), distance / 2)."""
    base, queries = golden_inputs(oracle, "g1_unit_sqeuclidean")
    d, i = oracle.knn_canonical(base, queries, 10, "sqeuclidean")
    for r in range(0, len(queries), 7):
        dots = base[i[r]].astype(np.float64) @ queries[r].astype(np.float64)
        assert np.isclose(1.0 - dots, d[r] / 2.0).all()


@pytest.mark.parametrize("name", ["basic", "ties", "single", "nine"])
def test_oracle_merge_matches_reference_merge(oracle, name):
    """Golden outputs came from running /root/reference/neighborhoodwatch/merge.py itself."""
    g = np.load(GOLDEN / f"merge_{name}.npz")
    k = int(g["k"])
    d, i = oracle.merge_partials(g["in_dist"], g["in_idx"], k)
    assert str(g["out_dist_dtype"]) == "float32" and str(g["out_idx_dtype"]) == "int64"
    assert list(g["out_columns"]) == [str(c) for c in range(k)]
    assert np.array_equal(d.view(np.uint32), g["out_dist"].view(np.uint32)), "merged distances differ from the reference"
    # ids: identical wherever the distance is unique in its row; inside a run of equal distances the reference's
    # order is numpy's unstable argsort, so compare such runs as sets
    ref_i = g["out_idx"]
    for r in range(d.shape[0]):
        for v in np.unique(d[r]):
            m = d[r] == v
            if d[r, -1] == v:       # run may be cut by the k boundary: any subset of the tied ids is legitimate
                pool = set(g["in_idx"][:, r, :][g["in_dist"][:, r, :] == v].tolist())
                assert set(ref_i[r, m].tolist()) <= pool and set(i[r, m].tolist()) <= pool
            else:
                assert set(i[r, m].tolist()) == set(ref_i[r, m].tolist())
    f_d, f_i = oracle.merge_partials_numpy_fold(list(g["in_dist"]), list(g["in_idx"]), k)
    assert np.array_equal(f_d, g["out_dist"])


def test_merge_ties_are_broken_by_lower_index(oracle):
    d = np.array([[[1.0, 2.0]], [[1.0, 2.0]]], dtype=np.float32)
    i = np.array([[[9, 4]], [[3, 8]]], dtype=np.int64)
    od, oi = oracle.merge_partials(d, i, 3)
    assert od.tolist() == [[1.0, 1.0, 2.0]] and oi.tolist() == [[3, 9, 4]]


def test_generators_agree(oracle):
    assert np.array_equal(oracle.gen(123, 7, 50, 33), oracle.gen_numpy(123, 7, 50, 33))
    x = oracle.gen(oracle.SEED_BASE, 0, 2000, 64)
    assert x.min() >= -1 and x.max() < 1 and abs(float(x.mean())) < 0.01
    assert np.array_equal(x * np.float32(2 ** 23), np.round(x * np.float32(2 ** 23)))      # on the 2^-23 grid


def test_oracle_rejects_k_larger_than_rows(oracle):
    with pytest.raises(ValueError):
        oracle.knn_canonical(oracle.gen(1, 0, 5, 8), oracle.gen(2, 0, 2, 8), 6)
