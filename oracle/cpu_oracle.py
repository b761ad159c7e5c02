"""CPU oracle for the brute-force kNN hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker.  The product (``neighborhoodwatch_amd``) never imports it.

Three layers, each citing what it restates (paths relative to /root/reference):

* :func:`knn_canonical` -- ctypes binding of ``oracle/nwknn_oracle.c``: the bit-exact restatement of
  the operator at ``neighborhoodwatch/cu_knn.py:243-245`` under the canonical arithmetic of
  DESIGN.md (k-ascending fmaf chains).  kNN parity is unpinned by the reference's own tests
  (``tests/test_02_cu_knn.py`` asserts nothing); see the C file header.
* :func:`knn_exact_fp64` -- float64 brute force, ranked by (distance, index): the independent
  mathematical truth every fp32 result is compared against (indices modulo fp32 near-ties,
  distances within 1e-5 relative to the distance scale).
* :func:`knn_blas_fp32` -- numpy expanded form ``|q|^2 + |b|^2 - 2 q.b^T`` with sgemm +
  argpartition: what a CPU user of sklearn/numpy would run; used as the timed CPU baseline.
* :func:`merge_partials` -- restatement of ``neighborhoodwatch/merge.py:56-101``.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libnwknn_oracle.so"
_lib = None

METRICS = {"sqeuclidean": 0, "one_minus_dot": 1, "cosine": 2}


def build(force: bool = False) -> Path:
    """Compile oracle/nwknn_oracle.c with the committed Makefile (gcc only)."""
    src = _HERE / "nwknn_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)
    return _LIB_PATH


def _usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (the GPU boxes show 256 CPUs and grant 16 -- OpenMP's
    default of one thread per CPU shown made the oracle 25x slower there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = ctypes.CDLL(str(_LIB_PATH))
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.nwo_version.restype = ctypes.c_int
        L.nwo_gen.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, f32p]
        L.nwo_gen.restype = None
        L.nwo_row_norms.argtypes = [f32p, ctypes.c_int64, ctypes.c_int32, f32p]
        L.nwo_row_norms.restype = None
        for name in ("nwo_knn_scalar",):
            fn = getattr(L, name)
            fn.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                           ctypes.c_int, f32p, i64p]
            fn.restype = ctypes.c_int
        L.nwo_knn.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                              ctypes.c_int, f32p, i64p, ctypes.c_int]
        L.nwo_knn.restype = ctypes.c_int
        L.nwo_merge.argtypes = [f32p, i64p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                f32p, i64p]
        L.nwo_merge.restype = ctypes.c_int
        L.nwo_maxsim.argtypes = [f32p, ctypes.c_int64, ctypes.c_int32, f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, f32p, i64p]
        L.nwo_maxsim.restype = ctypes.c_int
        L.nwo_distance.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.nwo_distance.restype = ctypes.c_float
        L.nwo_sortable_u32.argtypes = [ctypes.c_float]
        L.nwo_sortable_u32.restype = ctypes.c_uint32
        L.nwo_has_avx2_fma.restype = ctypes.c_int
        L.nwo_set_threads.argtypes = [ctypes.c_int]
        L.nwo_get_threads.restype = ctypes.c_int
        L.nwo_set_threads(_usable_cpus())
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


# ---------------------------------------------------------------------------------------------
# synthetic inputs: counter-based, integer-exact, identical on host and device
# ---------------------------------------------------------------------------------------------
SEED_BASE = 0x6E770001
SEED_QUERY = 0x6E770002
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G1 = np.uint64(0x9E3779B97F4A7C15)
_G2 = np.uint64(0xD1342543DE82EF95)


def _fmix64(z: np.ndarray) -> np.ndarray:
    z = z ^ (z >> np.uint64(30)); z = z * _M1
    z = z ^ (z >> np.uint64(27)); z = z * _M2
    return z ^ (z >> np.uint64(31))


def gen_numpy(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """Pure-numpy statement of the generator (the definition; the C and HIP versions must equal it)."""
    with np.errstate(over="ignore"):
        r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        c = (np.arange(d, dtype=np.uint64) + np.uint64(1))[None, :]
        rk = _fmix64(np.uint64(seed) + r * _G1)
        u = _fmix64(rk + c * _G2)
    v = (u >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (v.astype(np.float32) * np.float32(1.0 / 8388608.0)).astype(np.float32)


def gen(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    out = np.empty((rows, d), dtype=np.float32)
    lib().nwo_gen(ctypes.c_uint64(seed), row0, rows, d, _fp(out))
    return out


def gen_f16_numpy(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """fp16 variant of the generator: top 11 bits of the same 64-bit draw, signed, times 2^-10 (exact in fp16)."""
    with np.errstate(over="ignore"):
        r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        c = (np.arange(d, dtype=np.uint64) + np.uint64(1))[None, :]
        u = _fmix64(_fmix64(np.uint64(seed) + r * _G1) + c * _G2)
    v = (u >> np.uint64(53)).astype(np.int64) - (1 << 10)
    return (v.astype(np.float32) * np.float32(1.0 / 1024.0)).astype(np.float16)


def gen_unit_rows(seed: int, row0: int, rows: int, d: int) -> np.ndarray:
    """Seeded stand-in for the reference's test data (tests/test_02_cu_knn.py:15-28: np.random.rand rows,
    each L2-normalised by nw_utils.normalize_vector, nw_utils.py:56-59): values on a 2^-24 grid in [0,1),
    norm from an exact int64 sum of squares, one float64 divide, one cast -- machine independent."""
    with np.errstate(over="ignore"):
        r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        c = (np.arange(d, dtype=np.uint64) + np.uint64(1))[None, :]
        u = _fmix64(_fmix64(np.uint64(seed) + r * _G1) + c * _G2)
    v = (u >> np.uint64(40)).astype(np.int64)                      # [0, 2^24)
    ss = (v * v).sum(axis=1, keepdims=True)                        # exact (< 2^63 for d < 2^15)
    return (v.astype(np.float64) / np.sqrt(ss.astype(np.float64))).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# oracles
# ---------------------------------------------------------------------------------------------
def row_norms(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().nwo_row_norms(_fp(x), x.shape[0], x.shape[1], _fp(out))
    return out


def knn_canonical(dataset, queries, k: int, metric: str = "sqeuclidean", threads: int = 0, scalar: bool = False):
    """(distances f32[q,k], indices i64[q,k]) -- same positional order as pylibraft's knn(dataset, queries, k)
    at cu_knn.py:243-245."""
    b, q = _f32(dataset), _f32(queries)
    if b.ndim != 2 or q.ndim != 2 or b.shape[1] != q.shape[1]:
        raise ValueError("dataset and queries must be 2-D with equal dimension")
    if not (0 < k <= b.shape[0]):
        raise ValueError(f"k={k} must satisfy 0 < k <= {b.shape[0]} rows")   # cu_knn.py:237
    od = np.empty((q.shape[0], k), dtype=np.float32)
    oi = np.empty((q.shape[0], k), dtype=np.int64)
    if scalar:
        rc = lib().nwo_knn_scalar(_fp(b), b.shape[0], _fp(q), q.shape[0], b.shape[1], k, METRICS[metric], _fp(od), _ip(oi))
    else:
        rc = lib().nwo_knn(_fp(b), b.shape[0], _fp(q), q.shape[0], b.shape[1], k, METRICS[metric], _fp(od), _ip(oi), threads)
    if rc != 0:
        raise RuntimeError(f"oracle failed rc={rc}")
    return od, oi


def distances_fp64(dataset, queries, metric: str = "sqeuclidean") -> np.ndarray:
    b = np.asarray(dataset, dtype=np.float64)
    q = np.asarray(queries, dtype=np.float64)
    dot = q @ b.T
    if metric == "sqeuclidean":
        d = (q * q).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * dot
        return np.maximum(d, 0.0)
    if metric == "one_minus_dot":
        return 1.0 - dot
    if metric == "cosine":
        return 1.0 - dot / (np.linalg.norm(q, axis=1)[:, None] * np.linalg.norm(b, axis=1)[None, :])
    raise ValueError(metric)


def knn_exact_fp64(dataset, queries, k: int, metric: str = "sqeuclidean"):
    d = distances_fp64(dataset, queries, metric)
    order = np.lexsort((np.broadcast_to(np.arange(d.shape[1]), d.shape), d), axis=1)[:, :k]
    return np.take_along_axis(d, order, axis=1), order.astype(np.int64)


def knn_blas_fp32(dataset, queries, k: int, metric: str = "sqeuclidean", tile: int = 131072):
    """numpy/BLAS brute force in base tiles, as a CPU user would write it."""
    b, q = _f32(dataset), _f32(queries)
    qn = np.einsum("ij,ij->i", q, q)
    best_d = best_i = None
    for lo in range(0, b.shape[0], tile):
        bt = b[lo:lo + tile]
        dot = q @ bt.T
        if metric == "sqeuclidean":
            dist = qn[:, None] + np.einsum("ij,ij->i", bt, bt)[None, :] - 2.0 * dot
            np.maximum(dist, 0.0, out=dist)
        elif metric == "one_minus_dot":
            dist = 1.0 - dot
        else:
            dist = 1.0 - dot / (np.sqrt(qn)[:, None] * np.linalg.norm(bt, axis=1)[None, :])
        kk = min(k, bt.shape[0])
        part = np.argpartition(dist, kk - 1, axis=1)[:, :kk]
        pd_ = np.take_along_axis(dist, part, axis=1)
        pi_ = part.astype(np.int64) + lo
        if best_d is not None:
            pd_ = np.concatenate([best_d, pd_], axis=1)
            pi_ = np.concatenate([best_i, pi_], axis=1)
        order = np.lexsort((pi_, pd_), axis=1)[:, :k]
        best_d = np.take_along_axis(pd_, order, axis=1)
        best_i = np.take_along_axis(pi_, order, axis=1)
    return best_d.astype(np.float32), best_i


def merge_partials(dists, idx, k: int):
    """dists f32[n_lists,Q,k_in], idx i64[n_lists,Q,k_in] -> (f32[Q,k], i64[Q,k]) by (distance, index).
    Restates merge.py:56-101 (sequential fold of [Q,2k] argsorts == global k smallest)."""
    d = np.ascontiguousarray(dists, dtype=np.float32)
    i = np.ascontiguousarray(idx, dtype=np.int64)
    n, q, kin = d.shape
    od = np.empty((q, k), dtype=np.float32)
    oi = np.empty((q, k), dtype=np.int64)
    rc = lib().nwo_merge(_fp(d), _ip(i), n, q, kin, k, _fp(od), _ip(oi))
    if rc != 0:
        raise RuntimeError(f"oracle merge failed rc={rc}")
    return od, oi


def maxsim_canonical(doc_tokens, query_tokens, k: int):
    """MaxSim under the engine's canonical arithmetic (see nwo_maxsim): (scores f32 [q,k] descending, doc ids i64 [q,k])."""
    dt = np.ascontiguousarray(doc_tokens, dtype=np.float32); qt = np.ascontiguousarray(query_tokens, dtype=np.float32)
    nd, td, d = dt.shape; nq, tq, d2 = qt.shape
    assert d == d2
    os_ = np.empty((nq, k), dtype=np.float32); od = np.empty((nq, k), dtype=np.int64)
    rc = lib().nwo_maxsim(_fp(dt), nd, td, _fp(qt), nq, tq, d, k, _fp(os_), _ip(od))
    if rc != 0:
        raise RuntimeError(f"oracle maxsim failed rc={rc}")
    return os_, od


def maxsim_exact_fp64(doc_tokens, query_tokens, k: int):
    """float64 MaxSim, ranked by (-score, doc id): the independent truth."""
    dt = np.asarray(doc_tokens, dtype=np.float64); qt = np.asarray(query_tokens, dtype=np.float64)
    nd, td, d = dt.shape
    scores = np.empty((qt.shape[0], nd))
    flat = dt.reshape(nd * td, d)
    for q in range(qt.shape[0]):
        sim = (qt[q] @ flat.T).reshape(qt.shape[1], nd, td)       # [tq, nd, td]
        scores[q] = sim.max(axis=2).sum(axis=0)
    order = np.lexsort((np.broadcast_to(np.arange(nd), scores.shape), -scores), axis=1)[:, :k]
    return np.take_along_axis(scores, order, axis=1), order.astype(np.int64)


def merge_partials_numpy_fold(dists, idx, k: int):
    """The reference's fold, literally in numpy: concat [Q,2k], argsort rows, keep k (merge.py:73-101).
    np.argsort(kind default) is unstable, so equal distances may come out in either order."""
    fd, fi = np.asarray(dists[0]), np.asarray(idx[0])
    for j in range(1, len(dists)):
        cd = np.concatenate([fd, dists[j]], axis=1)
        ci = np.concatenate([fi, idx[j]], axis=1)
        o = np.argsort(cd, axis=1)
        fd = np.take_along_axis(cd, o, axis=1)[:, :k]
        fi = np.take_along_axis(ci, o, axis=1)[:, :k]
    return fd, fi


# ---------------------------------------------------------------------------------------------
# comparison helpers shared by the parity tests
# ---------------------------------------------------------------------------------------------
def pair_distances_fp64(dataset, queries, idx, metric: str = "sqeuclidean") -> np.ndarray:
    """float64 distance of every returned (query row, neighbour id) pair: [q,k]."""
    b = np.asarray(dataset, dtype=np.float64)
    q = np.asarray(queries, dtype=np.float64)
    idx = np.asarray(idx)
    out = np.empty(idx.shape, dtype=np.float64)
    for r in range(idx.shape[0]):
        nb = b[idx[r]]
        dot = nb @ q[r]
        if metric == "sqeuclidean":
            out[r] = np.maximum((q[r] * q[r]).sum() + (nb * nb).sum(1) - 2.0 * dot, 0.0)
        elif metric == "one_minus_dot":
            out[r] = 1.0 - dot
        else:
            out[r] = 1.0 - dot / (np.linalg.norm(q[r]) * np.linalg.norm(nb, axis=1))
    return out


def check_against_exact(dataset, queries, dist32, idx32, exact_dist_sorted, metric: str = "sqeuclidean",
                        rel_tol: float = 1e-5):
    """Near-tie-aware parity against the float64 truth (hard part 1).

    tolerance t = rel_tol * max(1, |q|^2 + |b|^2) per pair (1e-5 absolute for unit vectors, the bar
    BASELINE.json states).  Passes iff, for every query row:
      * ids are unique and in range;
      * |dist32 - d64(q, id)| <= t                 (reported distance is right for the reported id);
      * d64(q, id_j) <= exact_j + 2t               (the j-th reported neighbour is as good as the true j-th);
      * dist32 is ascending.
    Returns (max_abs_err, worst_rank_slack)."""
    dist32 = np.asarray(dist32); idx32 = np.asarray(idx32)
    b = np.asarray(dataset, dtype=np.float64); q = np.asarray(queries, dtype=np.float64)
    assert idx32.min() >= 0 and idx32.max() < b.shape[0], "neighbour id out of range"
    srt = np.sort(idx32, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate neighbour ids in a row"
    assert (np.diff(dist32.astype(np.float64), axis=1) >= 0).all(), "distances not ascending"   # merge.py:104
    d64 = pair_distances_fp64(b, q, idx32, metric)
    bn = (b * b).sum(1)
    t = rel_tol * np.maximum(1.0, (q * q).sum(1)[:, None] + bn[idx32])
    err = np.abs(dist32.astype(np.float64) - d64)
    assert (err <= t).all(), f"distance error {err.max():.3e} exceeds tolerance"
    slack = d64 - np.asarray(exact_dist_sorted)
    assert (slack <= 2 * t).all(), f"returned neighbour worse than the true one by {slack.max():.3e}"
    return float(err.max()), float(slack.max())
