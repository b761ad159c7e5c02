"""Python face of the HIP engine: a thin, typed wrapper over the C ABI.

``knn(dataset, queries, k)`` keeps the positional order and the ``(distances, indices)`` return tuple of
``pylibraft.neighbors.brute_force.knn`` as the reference calls it (cu_knn.py:243-245); ``KnnEngine`` is the
persistent form (upload a base shard once, search many query sets) that ``process_batches`` and the
multi-GPU driver use.  numpy arrays are host buffers; torch CUDA tensors are passed by device pointer
(torch is plumbing for memory and streams only -- no torch op touches the data).
"""
from __future__ import annotations

import atexit
import ctypes
import sys
import weakref
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import DEVICE, DTYPE_F16, DTYPE_F32, DTYPE_F64, ENGINE_TO_METRIC, HOST, METRIC_IDS, NwknnStats, check


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def resolve_metric(metric: Optional[str] = None, engine: Optional[str] = None) -> str:
    """`engine` values of the reference ('raft' | 'cuvs' | 'torch') are accepted as metric aliases."""
    if metric is None:
        metric = ENGINE_TO_METRIC.get(engine or "raft")
        if metric is None:
            raise ValueError(f"unknown engine {engine!r}; expected one of {sorted(ENGINE_TO_METRIC)}")
    if metric not in METRIC_IDS:
        raise ValueError(f"unknown metric {metric!r}; expected one of {sorted(METRIC_IDS)}")
    return metric


# Handles still open when the interpreter exits are destroyed HERE, in an atexit hook -- the first thing Py_FinalizeEx runs, with every
# module, torch's allocator and the HIP runtime intact -- and never from a __del__ that runs during interpreter finalization.
# Why (DESIGN.md section 9 has the reading of the code this rests on): round 3 saw ONE process exit on a signal after a green GPU test run
# and kept no log of it, so the trigger is not known.  nwknn_destroy is nine kinds of HIP call (set device, stream synchronise, hipFree x 26,
# event / stream destroy, hipHostFree).  Called from __del__ during finalization it runs after torch's Python side has been torn down in
# an order nobody controls (module dicts are cleared, then whatever objects remain are freed) -- the runtime library itself is still
# loaded then, its own exit handlers run later, in exit() -- so the atexit hook removes the only window in which OUR calls run in an
# unspecified state; it cannot fix a fault that is not ours.  What makes a recurrence readable: __del__ says so on stderr when it meets an
# open handle at finalization (it must not: _OPEN_AT_FINALIZATION), the exit test runs its child under faulthandler and keeps the child's
# stderr and exit status in gpurun_out/ when they are not clean.
_live = weakref.WeakSet()
_OPEN_AT_FINALIZATION = b"libnwknn: a handle was still open when the interpreter finalised; no HIP call is made from __del__ at that point\n"
_closed_at_exit = 0


@atexit.register
def _close_live_handles() -> None:
    global _closed_at_exit
    for obj in list(_live):
        try:
            if obj._is_open():
                obj.close()
                _closed_at_exit += 1
        except Exception:
            pass


def _del_guard(obj) -> bool:
    """True when __del__ may destroy the handle.  During interpreter finalization it may not (see above); an open handle there means the
    atexit hook missed it, which is reported, not papered over."""
    if not sys.is_finalizing():
        return True
    try:
        if obj._is_open():
            import os
            os.write(2, _OPEN_AT_FINALIZATION)
    except Exception:
        pass
    return False


class KnnEngine:
    """One GPU's share of a brute-force kNN job (the reference uses device 0 only, cu_knn.py:65)."""

    def __init__(self, device: int = 0, profile: bool = False):
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        check(self._L.nwknn_create(ctypes.byref(self._h), int(device)), "nwknn_create")
        _live.add(self)
        self.device = int(device)
        self.rows = 0
        self.dim = 0
        if profile:
            self.set_option("profile", 1)

    # -- lifetime ---------------------------------------------------------------------------------
    def _is_open(self) -> bool:
        return getattr(self, "_h", None) is not None and bool(self._h.value)

    def close(self) -> None:
        if self._is_open():
            self._L.nwknn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        if not _del_guard(self):                                    # too late for HIP calls (see _close_live_handles)
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- knobs ------------------------------------------------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        check(self._L.nwknn_set_option(self._h, name.encode(), int(value)), f"nwknn_set_option({name})")

    def set_stream(self, stream_ptr: int) -> None:
        check(self._L.nwknn_set_stream(self._h, ctypes.c_void_p(int(stream_ptr) or None)), "nwknn_set_stream")

    def synchronize(self) -> None:
        check(self._L.nwknn_synchronize(self._h), "nwknn_synchronize")

    def stats(self) -> dict:
        s = NwknnStats()
        check(self._L.nwknn_get_stats(self._h, ctypes.byref(s)), "nwknn_get_stats")
        return s.as_dict()

    def tile_kernel_name(self) -> str:
        """The distance-tile kernel behind the last search's filtered launches, as rocprofv3 spells it."""
        return (self._L.nwknn_tile_kernel_name(self._h) or b"").decode()

    # -- data -------------------------------------------------------------------------------------
    @staticmethod
    def _matrix(x, what: str):
        """-> (pointer, rows, dim, dtype id, location, keepalive).  float16 inputs stay float16 (fp16 compute path);
        everything else is taken as float32 (``dataset.astype(np.float32)``, cu_knn.py:243)."""
        if _is_torch(x):
            import torch
            if x.dtype not in (torch.float32, torch.float16):
                x = x.to(torch.float32)
            if x.dim() != 2:
                raise ValueError(f"{what} must be 2-D, got shape {tuple(x.shape)}")
            x = x.contiguous()
            loc = DEVICE if x.is_cuda else HOST
            return x.data_ptr(), x.shape[0], x.shape[1], (DTYPE_F16 if x.dtype == torch.float16 else DTYPE_F32), loc, x
        a = np.asarray(x)
        a = np.ascontiguousarray(a, dtype=np.float16 if a.dtype == np.float16 else np.float32)
        if a.ndim != 2:
            raise ValueError(f"{what} must be 2-D, got shape {a.shape}")
        return a.ctypes.data, a.shape[0], a.shape[1], (DTYPE_F16 if a.dtype == np.float16 else DTYPE_F32), HOST, a

    def _order_after_torch(self, t, what: str) -> None:
        """A CUDA tensor handed to the library: it must live on this engine's GPU, and whatever torch queued on its own
        stream to produce it (``.contiguous()``, ``.to(float32)``, the caller's kernels) must have finished before the
        engine's stream reads the pointer."""
        import torch
        if t.device.index != self.device:
            raise ValueError(f"{what} lives on {t.device}, this engine drives cuda:{self.device}")
        torch.cuda.current_stream(t.device).synchronize()

    def set_compute_dtype(self, dtype: str) -> None:
        """'f32' (default, bit-exact canonical arithmetic) or 'f16' (fp16 MFMA, fp32 accumulate; BASELINE config 4).
        Takes effect at the next set_base."""
        self.set_option("compute_dtype", {"f32": DTYPE_F32, "fp32": DTYPE_F32, "f16": DTYPE_F16, "fp16": DTYPE_F16}[dtype])

    def set_prescreen(self, on: bool) -> None:
        """fp32 searches (all three metrics) run an fp16 MFMA pre-screen over scaled fp16 copies of both sides and
        rescore every listed pair with the canonical fp32 chain: the SAME bits out as the direct fp32 kernel, several times
        faster (DESIGN.md "pre-screen").  Costs an extra fp16 copy of the base in HBM.  Takes effect at the next set_base."""
        self.set_option("prescreen", 1 if on else 0)

    def set_base(self, dataset) -> None:
        ptr, rows, dim, dt, loc, keep = self._matrix(dataset, "dataset")
        if rows == 0 or dim == 0:
            raise ValueError("dataset is empty")
        if loc == DEVICE:
            self._order_after_torch(keep, "dataset")
        check(self._L.nwknn_set_base(self._h, ctypes.c_void_p(ptr), rows, dim, dt, loc), "nwknn_set_base")
        if loc == DEVICE:
            self.synchronize()               # the caller may free/overwrite its tensor right after we return
        self.rows, self.dim = rows, dim
        del keep

    # -- column-major input (what parquet / arrow hold: one array per dimension) ------------------------------
    @staticmethod
    def _columns(columns, what: str):
        """-> (ctypes pointer table [chunk][dim], ctypes int64 rows per chunk, n_chunks, rows, dim, dtype id, keepalive list).
        Every column is a 1-D array, or a list of 1-D arrays (the chunks of a pyarrow ChunkedArray: one per parquet row group);
        all columns share one dtype (float32 | float64 | float16) and one chunking.  Nothing is concatenated here."""
        def chunks_of(c):
            # a list/tuple whose elements are themselves 1-D array-likes is a list of chunks; anything else (an array, a plain
            # Python list of numbers) is ONE chunk
            if isinstance(c, (list, tuple)) and len(c) > 0 and all(np.ndim(x) == 1 for x in c):
                return [np.asarray(x) for x in c]
            return [np.asarray(c)]
        cols = [chunks_of(c) for c in columns]
        if not cols:
            raise ValueError(f"{what}: no columns")
        for j, c in enumerate(cols):
            for x in c:
                if x.ndim != 1:
                    raise ValueError(f"{what}: column {j} must be a 1-D array or a list of 1-D chunks, got shape {x.shape}")
        dt = cols[0][0].dtype if cols[0] else np.dtype(np.float32)
        if dt not in (np.float32, np.float64, np.float16):
            cols = [[x.astype(np.float32) for x in c] for c in cols]
            dt = np.dtype(np.float32)
        chunk_rows = [x.shape[0] for x in cols[0]]
        keep = []
        for j, c in enumerate(cols):
            if [x.shape[0] for x in c] != chunk_rows or any(x.ndim != 1 or x.dtype != dt for x in c):
                raise ValueError(f"{what}: column {j} has chunks {[(x.shape, str(x.dtype)) for x in c]}, expected rows {chunk_rows} of {dt}")
            keep.append([np.ascontiguousarray(x) for x in c])
        n_chunks, dim = len(chunk_rows), len(keep)
        table = (ctypes.c_void_p * (n_chunks * dim))(*[keep[j][c].ctypes.data for c in range(n_chunks) for j in range(dim)])
        rows_arr = (ctypes.c_int64 * n_chunks)(*chunk_rows)
        ident = {np.dtype(np.float32): DTYPE_F32, np.dtype(np.float64): DTYPE_F64, np.dtype(np.float16): DTYPE_F16}[np.dtype(dt)]
        return table, rows_arr, n_chunks, int(sum(chunk_rows)), dim, ident, keep

    def set_base_columns(self, columns) -> None:
        """Base shard from per-dimension host arrays (or lists of chunks): no row-major copy and no concatenation is made on the
        host or the device (the job of ``cp.from_dlpack(df.to_dlpack()).copy(order='C')``, cu_knn.py:211)."""
        table, rows_arr, n_chunks, rows, dim, dt, keep = self._columns(columns, "dataset")
        if rows == 0:
            raise ValueError("dataset is empty")
        check(self._L.nwknn_set_base_column_chunks(self._h, table, rows_arr, n_chunks, rows, dim, dt), "nwknn_set_base_column_chunks")
        self.rows, self.dim = rows, dim
        del keep

    # -- a base handed over in pieces (the row groups of a tile, as they are decoded) ---------------------------------------------------
    def base_begin(self, rows: int, dim: int, dtype: str = "f32") -> None:
        """Announce a base of ``rows`` x ``dim`` that will arrive through ``base_append*`` in row order (``dtype`` 'f16': the pieces are
        fp16 and the fp16 compute path is used, as ``set_base`` of a float16 matrix).  The previous base is gone from here on."""
        if rows <= 0 or dim <= 0:
            raise ValueError("dataset is empty")
        dt = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "f16": DTYPE_F16, "fp16": DTYPE_F16}[dtype]
        self.rows = 0
        self._attached = None
        check(self._L.nwknn_base_begin(self._h, int(rows), int(dim), dt), "nwknn_base_begin")
        self._pending = (int(rows), int(dim))

    def base_append(self, rows) -> None:
        """The next rows as a row-major matrix (numpy, or a torch tensor on the host or on this engine's GPU)."""
        ptr, n, dim, dt, loc, keep = self._matrix(rows, "piece")
        if loc == DEVICE:
            self._order_after_torch(keep, "piece")
        if dim != self._pending[1]:
            raise ValueError(f"piece has dimension {dim}, the announced base has {self._pending[1]}")
        check(self._L.nwknn_base_append(self._h, ctypes.c_void_p(ptr), n, dt, loc), "nwknn_base_append")
        if loc == DEVICE:
            self.synchronize()
        del keep

    @classmethod
    def prepare_columns(cls, columns):
        """The pointer table ``base_append_columns`` hands to the library, made ahead of time (a reader thread does it next to the decode:
        1536 columns are 1536 numpy views and ctypes pointers per piece, milliseconds the uploading thread does not have to spend)."""
        return ("prepared",) + cls._columns(columns, "piece")

    def base_append_columns(self, columns) -> None:
        """The next rows as per-dimension arrays (or lists of chunks), like ``set_base_columns`` -- or the result of ``prepare_columns``.
        Returns as soon as the arrays have been gathered into pinned memory: the copy to the GPU and the layout kernel overlap whatever
        the caller does next."""
        if isinstance(columns, tuple) and columns and isinstance(columns[0], str) and columns[0] == "prepared":
            table, rows_arr, n_chunks, n, dim, dt, keep = columns[1:]
        else:
            table, rows_arr, n_chunks, n, dim, dt, keep = self._columns(columns, "piece")
        if dim != self._pending[1]:
            raise ValueError(f"piece has dimension {dim}, the announced base has {self._pending[1]}")
        check(self._L.nwknn_base_append_column_chunks(self._h, table, rows_arr, n_chunks, n, dt), "nwknn_base_append_column_chunks")
        del keep

    def base_end(self) -> None:
        check(self._L.nwknn_base_end(self._h), "nwknn_base_end")
        self.rows, self.dim = self._pending

    # -- a search that walks the base while it arrives (nwknn_search_attach / nwknn_search_collect) --------------------------------------
    def search_attach(self, queries, k: int, metric: str = "sqeuclidean", index_offset: int = 0) -> bool:
        """Between ``base_begin`` and the first piece: start a search that launches every chunk of its walk as soon as the pieces covering
        it have been appended.  ``queries``: a row-major matrix, or ``("columns", per-dimension arrays)`` / ``("matrix", array)`` as
        ``cu_knn.table_input`` yields.  -> True when the search is attached (``search_collect`` after ``base_end`` returns its result,
        the bits ``search`` would give); False when the configured path needs the whole base first (fp16 compute path, pre-screen):
        call ``search`` after ``base_end`` as usual."""
        metric = resolve_metric(metric)
        if not getattr(self, "_pending", None):
            raise _lib.NwknnError("search_attach before base_begin")
        rows, dim = self._pending
        kind, data = queries if isinstance(queries, tuple) and len(queries) == 2 and isinstance(queries[0], str) else ("matrix", queries)
        attached = ctypes.c_int(0)
        if kind == "columns":
            table, rows_arr, n_chunks, nq, qdim, dt, keep = self._columns(data, "queries")
        else:
            ptr, nq, qdim, dt, loc, keep = self._matrix(data, "queries")
        if qdim != dim:
            raise ValueError(f"queries have dimension {qdim}, dataset has {dim}")
        if not (0 < int(k) <= rows):
            raise ValueError(f"k={k} must satisfy 0 < k <= {rows} dataset rows")           # cu_knn.py:237
        if nq == 0:
            raise ValueError("queries are empty")
        if kind == "columns":
            check(self._L.nwknn_search_attach_column_chunks(self._h, table, rows_arr, n_chunks, nq, dt, int(k), METRIC_IDS[metric], int(index_offset),
                                                            ctypes.byref(attached)), "nwknn_search_attach_column_chunks")
        else:
            if loc == DEVICE:
                self._order_after_torch(keep, "queries")
            check(self._L.nwknn_search_attach(self._h, ctypes.c_void_p(ptr), nq, dt, loc, int(k), METRIC_IDS[metric], int(index_offset),
                                              ctypes.byref(attached)), "nwknn_search_attach")
        del keep
        self._attached = (nq, int(k)) if attached.value else None
        return bool(attached.value)

    @property
    def search_attached(self) -> bool:
        return bool(getattr(self, "_attached", None))

    def search_collect(self, out_device: bool = False):
        """After ``base_end``: the attached search's (distances [q,k] f32, indices [q,k] i64)."""
        if not getattr(self, "_attached", None):
            raise _lib.NwknnError("search_collect without an attached search")
        nq, k = self._attached
        self._attached = None
        if out_device:
            import torch
            dev = torch.device("cuda", self.device)
            od = torch.empty((nq, k), dtype=torch.float32, device=dev)
            oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            pd_, pi_, oloc = od.data_ptr(), oi.data_ptr(), DEVICE
        else:
            od = np.empty((nq, k), dtype=np.float32)
            oi = np.empty((nq, k), dtype=np.int64)
            pd_, pi_, oloc = od.ctypes.data, oi.ctypes.data, HOST
        check(self._L.nwknn_search_collect(self._h, ctypes.c_void_p(pd_), ctypes.c_void_p(pi_), oloc), "nwknn_search_collect")
        return od, oi

    def search_columns(self, columns, k: int, metric: str = "sqeuclidean", index_offset: int = 0, out_device: bool = False):
        """Queries as per-dimension host arrays (or chunk lists).  -> (distances [q,k] f32, indices [q,k] i64): numpy arrays, or torch
        CUDA tensors on this engine's GPU with ``out_device=True`` (results that are folded on the device never visit the host)."""
        metric = resolve_metric(metric)
        table, rows_arr, n_chunks, nq, dim, dt, keep = self._columns(columns, "queries")
        if self.rows == 0:
            raise _lib.NwknnError("search before set_base")
        if dim != self.dim:
            raise ValueError(f"queries have dimension {dim}, dataset has {self.dim}")
        if not (0 < int(k) <= self.rows):
            raise ValueError(f"k={k} must satisfy 0 < k <= {self.rows} dataset rows")      # cu_knn.py:237
        if nq == 0:
            raise ValueError("queries are empty")
        if out_device:
            import torch
            dev = torch.device("cuda", self.device)
            od = torch.empty((nq, k), dtype=torch.float32, device=dev)
            oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            pd_, pi_, oloc = od.data_ptr(), oi.data_ptr(), DEVICE
        else:
            od = np.empty((nq, k), dtype=np.float32)
            oi = np.empty((nq, k), dtype=np.int64)
            pd_, pi_, oloc = od.ctypes.data, oi.ctypes.data, HOST
        check(self._L.nwknn_search_column_chunks(self._h, table, rows_arr, n_chunks, nq, dt, int(k), METRIC_IDS[metric], int(index_offset),
                                                 ctypes.c_void_p(pd_), ctypes.c_void_p(pi_), oloc), "nwknn_search_column_chunks")
        del keep
        return od, oi

    def search(self, queries, k: int, metric: str = "sqeuclidean", index_offset: int = 0, out_device: Optional[bool] = None):
        """-> (distances float32 [q,k], indices int64 [q,k]); torch CUDA tensors if the queries were, else numpy."""
        metric = resolve_metric(metric)
        ptr, nq, dim, dt, loc, keep = self._matrix(queries, "queries")
        if self.rows == 0:
            raise _lib.NwknnError("search before set_base")
        if dim != self.dim:
            raise ValueError(f"queries have dimension {dim}, dataset has {self.dim}")
        if not (0 < int(k) <= self.rows):
            raise ValueError(f"k={k} must satisfy 0 < k <= {self.rows} dataset rows")      # cu_knn.py:237
        if nq == 0:
            raise ValueError("queries are empty")
        want_dev = (loc == DEVICE) if out_device is None else bool(out_device)
        if loc == DEVICE:
            self._order_after_torch(keep, "queries")     # also when the results go to host memory
        if want_dev:
            import torch
            dev = torch.device("cuda", self.device)
            od = torch.empty((nq, k), dtype=torch.float32, device=dev)
            oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            pd_, pi_, oloc = od.data_ptr(), oi.data_ptr(), DEVICE
        else:
            od = np.empty((nq, k), dtype=np.float32)
            oi = np.empty((nq, k), dtype=np.int64)
            pd_, pi_, oloc = od.ctypes.data, oi.ctypes.data, HOST
        check(self._L.nwknn_search(self._h, ctypes.c_void_p(ptr), nq, dt, loc, int(k), METRIC_IDS[metric], int(index_offset),
                                   ctypes.c_void_p(pd_), ctypes.c_void_p(pi_), oloc), "nwknn_search")
        del keep
        return od, oi

    def pair_distances(self, queries, idx, metric: str = "sqeuclidean", index_offset: int = 0):
        """Canonical distances of GIVEN pairs: out[q, j] = distance(queries[q], dataset[idx[q, j] - index_offset]) -- the same bits a search
        would report for that pair; NaN where the row does not exist.  numpy in -> numpy out, torch CUDA in -> torch CUDA out.  This is
        what the reference's validator recomputes on the CPU, neighbour by neighbour (parquet_to_format.py:351-383)."""
        metric = resolve_metric(metric)
        ptr, nq, dim, dt, loc, keep = self._matrix(queries, "queries")
        if self.rows == 0:
            raise _lib.NwknnError("pair_distances before set_base")
        if dim != self.dim:
            raise ValueError(f"queries have dimension {dim}, dataset has {self.dim}")
        if loc == DEVICE:
            import torch
            self._order_after_torch(keep, "queries")
            it = idx.to(device=keep.device, dtype=torch.int64).contiguous()
            if it.dim() != 2 or it.shape[0] != nq:
                raise ValueError(f"idx must be [{nq}, k], got {tuple(it.shape)}")
            out = torch.empty(tuple(it.shape), dtype=torch.float32, device=keep.device)
            torch.cuda.current_stream(keep.device).synchronize()
            ip, op, k = it.data_ptr(), out.data_ptr(), it.shape[1]
        else:
            it = np.ascontiguousarray(idx, dtype=np.int64)
            if it.ndim != 2 or it.shape[0] != nq:
                raise ValueError(f"idx must be [{nq}, k], got {it.shape}")
            out = np.empty(it.shape, dtype=np.float32)
            ip, op, k = it.ctypes.data, out.ctypes.data, it.shape[1]
        check(self._L.nwknn_pair_distances(self._h, ctypes.c_void_p(ptr), nq, dt, loc, ctypes.c_void_p(ip), int(k), METRIC_IDS[metric], int(index_offset),
                                           ctypes.c_void_p(op), loc), "nwknn_pair_distances")
        del keep, it
        return out

    def pair_dots(self, queries, idx, index_offset: int = 0) -> np.ndarray:
        """float64 dot products of GIVEN pairs (host numpy in and out): out[q, j] = sum_k queries[q, k] * dataset[idx[q, j] - index_offset, k] in
        double precision, NaN where the row does not exist -- the arithmetic of the reference's validator (nwknn_pair_dots)."""
        ptr, nq, dim, dt, loc, keep = self._matrix(np.asarray(queries), "queries")
        if self.rows == 0:
            raise _lib.NwknnError("pair_dots before set_base")
        if dim != self.dim:
            raise ValueError(f"queries have dimension {dim}, dataset has {self.dim}")
        it = np.ascontiguousarray(idx, dtype=np.int64)
        if it.ndim != 2 or it.shape[0] != nq:
            raise ValueError(f"idx must be [{nq}, k], got {it.shape}")
        out = np.empty(it.shape, dtype=np.float64)
        check(self._L.nwknn_pair_dots(self._h, ctypes.c_void_p(ptr), nq, dt, HOST, ctypes.c_void_p(it.ctypes.data), int(it.shape[1]), int(index_offset),
                                      ctypes.c_void_p(out.ctypes.data), HOST), "nwknn_pair_dots")
        del keep
        return out

    MERGE_MAX_ENTRIES = 4096        # nwknn_merge sorts at most this many (list, rank) entries per query in LDS

    def merge(self, dists, idx, k: int):
        """dists/idx: [n_lists, Q, k_in] -> top-k by (distance, index), [Q, k] (merge.py:56-101 on the GPU).  More than
        4096 entries per query (e.g. 8 shards x k=1000) are folded in groups, which gives the same result."""
        torch_in = _is_torch(dists)
        if torch_in:
            import torch
            d = dists.contiguous().to(torch.float32); i = idx.contiguous().to(torch.int64)
        else:
            d = np.ascontiguousarray(dists, dtype=np.float32); i = np.ascontiguousarray(idx, dtype=np.int64)
        while True:
            n, q, kin = d.shape
            group = max(1, self.MERGE_MAX_ENTRIES // kin)
            if n <= group:
                return self._merge_once(d, i, int(k))
            keep = min(kin, int(k))
            parts = [self._merge_once(d[g0:g0 + group], i[g0:g0 + group], keep) for g0 in range(0, n, group)]
            if torch_in:
                import torch
                d = torch.stack([p[0] for p in parts]); i = torch.stack([p[1] for p in parts])
            else:
                d = np.stack([p[0] for p in parts]); i = np.stack([p[1] for p in parts])

    def _merge_once(self, d, i, k: int):
        n, q, kin = d.shape
        if n * kin > self.MERGE_MAX_ENTRIES:
            raise ValueError(f"k_in={kin} alone exceeds the {self.MERGE_MAX_ENTRIES}-entry merge limit")
        if _is_torch(d):
            import torch
            d = d.contiguous(); i = i.contiguous()
            od = torch.empty((q, k), dtype=torch.float32, device=d.device)
            oi = torch.empty((q, k), dtype=torch.int64, device=d.device)
            loc = DEVICE if d.is_cuda else HOST
            if d.is_cuda:
                self._order_after_torch(d, "merge input")
            check(self._L.nwknn_merge(self._h, ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(i.data_ptr()), n, q, kin, int(k),
                                      ctypes.c_void_p(od.data_ptr()), ctypes.c_void_p(oi.data_ptr()), loc), "nwknn_merge")
            return od, oi
        d = np.ascontiguousarray(d); i = np.ascontiguousarray(i)
        od = np.empty((q, k), dtype=np.float32); oi = np.empty((q, k), dtype=np.int64)
        check(self._L.nwknn_merge(self._h, ctypes.c_void_p(d.ctypes.data), ctypes.c_void_p(i.ctypes.data), n, q, kin, int(k),
                                  ctypes.c_void_p(od.ctypes.data), ctypes.c_void_p(oi.ctypes.data), HOST), "nwknn_merge")
        return od, oi

    # -- MaxSim late interaction (extension; not in the reference) -------------------------------------------
    MAXSIM_DOC_TOKENS = 64
    MAXSIM_QUERY_TOKENS = 32

    @staticmethod
    def pad_tokens(items, tokens: int, repeat_last: bool) -> np.ndarray:
        """Ragged token matrices (a sequence of [t_i, d] arrays, 1 <= t_i <= tokens; float16/32/64) -> dense float32 [n, tokens, d].
        Documents are padded by REPEATING their last token (a duplicate never changes max_j <q_i, d_j>); queries by ZERO tokens
        (max_j <0, d_j> = +0 adds nothing to the sum).  The kernels' tile shape fixes 64 tokens per document and 32 per query; this
        is how shorter ones are fed to them -- the score is then the canonical arithmetic on the padded matrices."""
        items = list(items)
        if not items:
            raise ValueError("no token matrices")
        d = int(np.asarray(items[0]).shape[-1])
        out = np.zeros((len(items), tokens, d), dtype=np.float32)
        for n, x in enumerate(items):
            a = np.asarray(x, dtype=np.float32)
            if a.ndim != 2 or a.shape[1] != d or not (1 <= a.shape[0] <= tokens):
                raise ValueError(f"item {n}: expected [1..{tokens}, {d}] tokens, got shape {a.shape}")
            out[n, :a.shape[0]] = a
            if repeat_last and a.shape[0] < tokens:
                out[n, a.shape[0]:] = a[-1]
        return out

    def maxsim_set_docs(self, doc_tokens) -> None:
        """doc_tokens: [n_docs, 64, d] float32 (numpy or torch CUDA), or a sequence of ragged [t_i <= 64, d] matrices (padded by
        repeating the last token, see pad_tokens)."""
        if isinstance(doc_tokens, (list, tuple)):
            doc_tokens = self.pad_tokens(doc_tokens, self.MAXSIM_DOC_TOKENS, repeat_last=True)
        if _is_torch(doc_tokens):
            import torch
            t = doc_tokens.to(torch.float32).contiguous()
            if t.dim() != 3:
                raise ValueError("doc_tokens must be [n_docs, tokens, d]")
            loc, ptr, shape, keep = (DEVICE if t.is_cuda else HOST), t.data_ptr(), tuple(t.shape), t
            if t.is_cuda:
                self._order_after_torch(t, "doc_tokens")
        else:
            a = np.ascontiguousarray(doc_tokens, dtype=np.float32)
            if a.ndim != 3:
                raise ValueError("doc_tokens must be [n_docs, tokens, d]")
            loc, ptr, shape, keep = HOST, a.ctypes.data, a.shape, a
        check(self._L.nwknn_maxsim_set_docs(self._h, ctypes.c_void_p(ptr), shape[0], shape[1], shape[2], DTYPE_F32, loc), "nwknn_maxsim_set_docs")
        if loc == DEVICE:
            self.synchronize()
        self.rows, self.dim, self.docs = shape[0] * shape[1], shape[2], shape[0]
        del keep

    def maxsim_search(self, query_tokens, k: int, doc_offset: int = 0):
        """query_tokens: [n_queries, 32, d] float32, or a sequence of ragged [t_i <= 32, d] matrices (zero-padded, see pad_tokens)
        -> (scores float32 [q,k] descending, doc ids int64 [q,k])."""
        if isinstance(query_tokens, (list, tuple)):
            query_tokens = self.pad_tokens(query_tokens, self.MAXSIM_QUERY_TOKENS, repeat_last=False)
        dev = _is_torch(query_tokens) and query_tokens.is_cuda
        if _is_torch(query_tokens):
            import torch
            t = query_tokens.to(torch.float32).contiguous()
            shape, ptr, keep, loc = tuple(t.shape), t.data_ptr(), t, (DEVICE if t.is_cuda else HOST)
            if t.is_cuda:
                self._order_after_torch(t, "query_tokens")
        else:
            a = np.ascontiguousarray(query_tokens, dtype=np.float32)
            shape, ptr, keep, loc = a.shape, a.ctypes.data, a, HOST
        if len(shape) != 3 or shape[2] != self.dim:
            raise ValueError(f"query_tokens must be [n_queries, tokens, {self.dim}]")
        nq = shape[0]
        if dev:
            import torch
            os_ = torch.empty((nq, k), dtype=torch.float32, device=keep.device)
            od = torch.empty((nq, k), dtype=torch.int64, device=keep.device)
            torch.cuda.current_stream(keep.device).synchronize()
            ps, pd_, oloc = os_.data_ptr(), od.data_ptr(), DEVICE
        else:
            os_ = np.empty((nq, k), dtype=np.float32); od = np.empty((nq, k), dtype=np.int64)
            ps, pd_, oloc = os_.ctypes.data, od.ctypes.data, HOST
        check(self._L.nwknn_maxsim_search(self._h, ctypes.c_void_p(ptr), nq, shape[1], DTYPE_F32, loc, int(k), int(doc_offset),
                                          ctypes.c_void_p(ps), ctypes.c_void_p(pd_), oloc), "nwknn_maxsim_search")
        del keep
        return os_, od

    def mfma_ceiling(self, dtype: str = "f16", seconds: float = 1.0) -> dict:
        """What a bare MFMA loop (random operands in registers, nothing else) delivers on this device: {'tflops', 'clock_ghz'}.  The
        power-limited ceiling a tile kernel of that dtype can be held against (nwknn_mfma_ceiling)."""
        tf, clk = ctypes.c_double(), ctypes.c_double()
        check(self._L.nwknn_mfma_ceiling(self._h, DTYPE_F16 if dtype in ("f16", "fp16") else DTYPE_F32, float(seconds), ctypes.byref(tf), ctypes.byref(clk)),
              "nwknn_mfma_ceiling")
        return {"tflops": tf.value, "clock_ghz": clk.value}

    def fabric_probe(self, nbytes: int = 128 << 20, seconds: float = 0.05) -> float:
        """GB/s a plain streaming read kernel gets over a working set of ``nbytes`` (128 MiB: beyond L2, inside the Infinity Cache; 4 GiB: HBM)."""
        out = ctypes.c_double()
        check(self._L.nwknn_fabric_probe(self._h, int(nbytes), float(seconds), ctypes.byref(out)), "nwknn_fabric_probe")
        return out.value

    def generate(self, seed: int, row0: int, rows: int, dim: int, dtype: str = "f32"):
        """Seeded synthetic rows straight into HBM (torch CUDA tensor [rows, dim], float32 or float16)."""
        import torch
        f16 = dtype in ("f16", "fp16")
        out = torch.empty((rows, dim), dtype=torch.float16 if f16 else torch.float32, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(out.device).synchronize()
        check(self._L.nwknn_generate(self._h, ctypes.c_uint64(seed), int(row0), int(rows), int(dim), DTYPE_F16 if f16 else DTYPE_F32,
                                     ctypes.c_void_p(out.data_ptr())), "nwknn_generate")
        self.synchronize()
        return out


class ShardedKnnEngine:
    """One process, N GPUs of one node (the C ABI's nwknn_comm_*): the base rows are sharded contiguously over ``devices``, queries are
    replicated, every GPU searches its shard with global row ids, ONE RCCL all-gather moves the [Q,k] partial lists over xGMI and GPU g
    merges its slice of the query rows by (distance, index).  The reference is pinned to GPU 0 (cu_knn.py:65); with one device this
    is ``KnnEngine`` (the exchange is a copy).  Inputs and outputs are host numpy arrays."""

    def __init__(self, devices=(0,), profile: bool = False):
        self._L = _lib.load()
        _lib.bind_to_torch_rccl()                                   # one RCCL per process: torch's copy, if torch is installed (see _lib.py)
        self._c = ctypes.c_void_p()
        devs = [int(x) for x in devices]
        arr = (ctypes.c_int * len(devs))(*devs)
        check(self._L.nwknn_comm_create(ctypes.byref(self._c), arr, len(devs)), "nwknn_comm_create")
        _live.add(self)
        self.devices = devs
        self.rows = self.dim = 0
        if profile:
            self.set_option("profile", 1)

    def _is_open(self) -> bool:
        return getattr(self, "_c", None) is not None and bool(self._c.value)

    def close(self) -> None:
        if self._is_open():
            self._L.nwknn_comm_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __del__(self):
        if not _del_guard(self):                                    # too late for HIP calls (see _close_live_handles)
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name: str, value: int) -> None:
        check(self._L.nwknn_comm_set_option(self._c, name.encode(), int(value)), f"nwknn_comm_set_option({name})")

    def set_prescreen(self, on: bool) -> None:
        self.set_option("prescreen", 1 if on else 0)

    def set_base(self, dataset) -> None:
        a = np.asarray(dataset)
        a = np.ascontiguousarray(a, dtype=np.float16 if a.dtype == np.float16 else np.float32)
        if a.ndim != 2 or a.shape[0] == 0 or a.shape[1] == 0:
            raise ValueError(f"dataset must be a non-empty 2-D array, got shape {a.shape}")
        check(self._L.nwknn_comm_set_base(self._c, ctypes.c_void_p(a.ctypes.data), a.shape[0], a.shape[1],
                                          DTYPE_F16 if a.dtype == np.float16 else DTYPE_F32), "nwknn_comm_set_base")
        self.rows, self.dim = a.shape

    def generate_base(self, seed: int, rows: int, dim: int, dtype: str = "f32") -> None:
        check(self._L.nwknn_comm_generate_base(self._c, ctypes.c_uint64(seed), int(rows), int(dim),
                                               DTYPE_F16 if dtype in ("f16", "fp16") else DTYPE_F32), "nwknn_comm_generate_base")
        self.rows, self.dim = int(rows), int(dim)

    def search(self, queries, k: int, metric: str = "sqeuclidean"):
        metric = resolve_metric(metric)
        q = np.asarray(queries)
        q = np.ascontiguousarray(q, dtype=np.float16 if q.dtype == np.float16 else np.float32)
        if q.ndim != 2 or q.shape[0] == 0:
            raise ValueError("queries must be a non-empty 2-D array")
        if self.rows == 0:
            raise _lib.NwknnError("search before set_base")
        if q.shape[1] != self.dim:
            raise ValueError(f"queries have dimension {q.shape[1]}, dataset has {self.dim}")
        od = np.empty((q.shape[0], k), dtype=np.float32)
        oi = np.empty((q.shape[0], k), dtype=np.int64)
        check(self._L.nwknn_search_sharded(self._c, ctypes.c_void_p(q.ctypes.data), q.shape[0], DTYPE_F16 if q.dtype == np.float16 else DTYPE_F32,
                                           int(k), METRIC_IDS[metric], ctypes.c_void_p(od.ctypes.data), ctypes.c_void_p(oi.ctypes.data)),
              "nwknn_search_sharded")
        return od, oi

    def maxsim_set_docs(self, doc_tokens) -> None:
        """Host doc tokens [n_docs, 64, d] float32 (or ragged matrices, padded as KnnEngine.maxsim_set_docs pads them), sharded contiguously
        over the devices."""
        if isinstance(doc_tokens, (list, tuple)):
            doc_tokens = KnnEngine.pad_tokens(doc_tokens, KnnEngine.MAXSIM_DOC_TOKENS, repeat_last=True)
        a = np.ascontiguousarray(doc_tokens, dtype=np.float32)
        if a.ndim != 3:
            raise ValueError("doc_tokens must be [n_docs, tokens, d]")
        check(self._L.nwknn_comm_maxsim_set_docs(self._c, ctypes.c_void_p(a.ctypes.data), a.shape[0], a.shape[1], a.shape[2], DTYPE_F32),
              "nwknn_comm_maxsim_set_docs")
        self.rows, self.dim, self.docs = a.shape[0] * a.shape[1], a.shape[2], a.shape[0]

    def maxsim_search(self, query_tokens, k: int):
        """-> (scores float32 [q,k] descending, global doc ids int64 [q,k]) over all the devices' documents."""
        if isinstance(query_tokens, (list, tuple)):
            query_tokens = KnnEngine.pad_tokens(query_tokens, KnnEngine.MAXSIM_QUERY_TOKENS, repeat_last=False)
        a = np.ascontiguousarray(query_tokens, dtype=np.float32)
        if a.ndim != 3 or a.shape[2] != self.dim:
            raise ValueError(f"query_tokens must be [n_queries, tokens, {self.dim}]")
        os_ = np.empty((a.shape[0], k), dtype=np.float32)
        od = np.empty((a.shape[0], k), dtype=np.int64)
        check(self._L.nwknn_maxsim_search_sharded(self._c, ctypes.c_void_p(a.ctypes.data), a.shape[0], a.shape[1], DTYPE_F32, int(k),
                                                  ctypes.c_void_p(os_.ctypes.data), ctypes.c_void_p(od.ctypes.data)), "nwknn_maxsim_search_sharded")
        return os_, od

    def merge_lists(self, dists, idx, k: int, n_slices: int = 0, descending: bool = False):
        """dists / idx: host [n_lists, Q, k] partial lists (idx < 0 = empty) -> ([Q,k], [Q,k]) by (distance, index): the merge step of a
        multi-device search (query rows in ``n_slices`` slices, default one per device; sliced and two-level merge) on lists the caller
        gathered -- merge.py:56-101 across any number of lists."""
        d = np.ascontiguousarray(dists, dtype=np.float32)
        i = np.ascontiguousarray(idx, dtype=np.int64)
        if d.ndim != 3 or d.shape != i.shape or d.shape[2] != int(k):
            raise ValueError(f"expected [n_lists, Q, {k}] distances and indices, got {d.shape} / {i.shape}")
        od = np.empty(d.shape[1:], dtype=np.float32)
        oi = np.empty(d.shape[1:], dtype=np.int64)
        check(self._L.nwknn_comm_merge_lists(self._c, ctypes.c_void_p(d.ctypes.data), ctypes.c_void_p(i.ctypes.data), d.shape[0], d.shape[1], int(k),
                                             int(n_slices) or len(self.devices), 1 if descending else 0, ctypes.c_void_p(od.ctypes.data), ctypes.c_void_p(oi.ctypes.data)),
              "nwknn_comm_merge_lists")
        return od, oi

    def stats(self) -> dict:
        s = _lib.NwknnCommStats()
        check(self._L.nwknn_comm_get_stats(self._c, ctypes.byref(s)), "nwknn_comm_get_stats")
        return s.as_dict()


_default_engines: dict = {}


def default_engine(device: int = 0) -> KnnEngine:
    eng = _default_engines.get(device)
    if eng is None:
        eng = _default_engines[device] = KnnEngine(device)
    return eng


def knn(dataset, queries, k: int, metric: Optional[str] = None, engine: Optional[str] = None, device: int = 0,
        prescreen: bool = False) -> Tuple:
    """Drop-in for ``pylibraft.neighbors.brute_force.knn(dataset, queries, k)`` as used at cu_knn.py:243-245:
    returns ``(distances float32 [q,k], indices int64 [q,k])``, ascending, indices local to ``dataset``."""
    eng = default_engine(device)
    eng.set_prescreen(prescreen)
    try:
        eng.set_base(dataset)
        return eng.search(queries, k, resolve_metric(metric, engine))
    finally:
        eng.set_prescreen(False)


def device_count() -> int:
    return int(_lib.load().nwknn_device_count())


def mem_info(device: int = 0) -> Tuple[int, int]:
    """(free, total) bytes of HBM -- stands in for the NVML query of tune_memory (cu_knn.py:62-71)."""
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    check(_lib.load().nwknn_mem_info(int(device), ctypes.byref(f), ctypes.byref(t)), "nwknn_mem_info")
    return int(f.value), int(t.value)
