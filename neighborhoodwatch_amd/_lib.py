"""ctypes binding of libnwknn.so (C ABI in include/nwknn.h).

The HIP library is the product: there is no CPU fallback.  Loading fails loudly when the shared
object is missing (run ``python -c 'import __graft_entry__ as g; g.build()'`` or
``python -m neighborhoodwatch_amd.build``), and creating a handle fails loudly without a gfx950 GPU.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from pathlib import Path

import os

PKG_DIR = Path(__file__).resolve().parent
# NWKNN_LIB_PATH: developers A/B two builds of the library inside one GPU session (tools/ab_libs.sh); unset in production
LIB_PATH = Path(os.environ["NWKNN_LIB_PATH"]) if os.environ.get("NWKNN_LIB_PATH") else PKG_DIR / "libnwknn.so"

NWKNN_OK = 0
HOST, DEVICE = 0, 1
DTYPE_F32, DTYPE_F16, DTYPE_F64 = 0, 1, 2
METRIC_IDS = {"sqeuclidean": 0, "one_minus_dot": 1, "cosine": 2}
# the reference selects the distance through its `engine` argument (cu_knn.py:193-200,243-262)
ENGINE_TO_METRIC = {"raft": "sqeuclidean", "torch": "one_minus_dot", "cuvs": "cosine", "hip": "sqeuclidean"}

EXPORTS = [
    "nwknn_version", "nwknn_last_error", "nwknn_device_count", "nwknn_mem_info", "nwknn_create", "nwknn_destroy",
    "nwknn_set_stream", "nwknn_set_option", "nwknn_get_stats", "nwknn_synchronize", "nwknn_set_base", "nwknn_search",
    "nwknn_merge", "nwknn_generate", "nwknn_set_base_columns", "nwknn_search_columns",
    "nwknn_set_base_column_chunks", "nwknn_search_column_chunks",
    "nwknn_base_begin", "nwknn_base_append", "nwknn_base_append_column_chunks", "nwknn_base_end",
    "nwknn_search_attach", "nwknn_search_attach_column_chunks", "nwknn_search_collect",
    "nwknn_maxsim_set_docs", "nwknn_maxsim_search", "nwknn_tile_kernel_name",
    "nwknn_comm_create", "nwknn_comm_destroy", "nwknn_comm_device_count", "nwknn_comm_engine", "nwknn_comm_set_option",
    "nwknn_comm_set_base", "nwknn_comm_generate_base", "nwknn_search_sharded", "nwknn_comm_get_stats", "nwknn_comm_merge_lists", "nwknn_mfma_ceiling", "nwknn_fabric_probe", "nwknn_pair_distances", "nwknn_pair_dots",
    "nwknn_comm_maxsim_set_docs", "nwknn_maxsim_search_sharded",
]


class NwknnStats(ctypes.Structure):
    _fields_ = [
        ("nq", c_int64), ("nb", c_int64), ("d", c_int32), ("k", c_int32),
        ("tile_launches", c_int32), ("compact_launches", c_int32), ("rescan_launches", c_int32),
        ("overflowed_queries", c_int32),
        ("tile_ms", c_double), ("compact_ms", c_double), ("prep_ms", c_double), ("total_ms", c_double),
        ("tile_flops", c_double), ("rescore_ms", c_double), ("prescreen", c_int32), ("prescreen_fallbacks", c_int32),
        ("prescreen_checked", c_int64), ("prescreen_violations", c_int64), ("prescreen_worst", c_double),
        ("walk_calibrated", c_int32), ("walk_query_fastest", c_int32), ("walk_ms_base_fastest", c_double), ("walk_ms_query_fastest", c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class NwknnCommStats(ctypes.Structure):
    _fields_ = [
        ("n_devices", c_int32), ("k", c_int32), ("nq", c_int64),
        ("search_ms", c_double), ("allgather_ms", c_double), ("merge_ms", c_double), ("allgather_bytes_per_device", c_double),
        ("tile_flops", c_double), ("tile_ms", c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class NwknnError(RuntimeError):
    pass


_lib = None


def _bind_to_torch_rocm_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (SONAME libamdhip64.so.7, the same as the
    system one).  A process must hold ONE HIP/HSA runtime: if libnwknn.so pulled in /opt/rocm's copy first, a later
    `import torch` would load a second runtime and fail with "No HIP GPUs are available", and device pointers could
    not be shared.  Loading torch's copy first (by path, RTLD_GLOBAL) makes libnwknn.so's NEEDED libamdhip64.so.7
    resolve to it, so torch tensors' data_ptr() and our streams/kernels live in the same runtime.  Without torch
    installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _torch_lib_dir():
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    return Path(list(spec.submodule_search_locations)[0]) / "lib"


def bind_to_torch_rccl() -> None:
    """One RCCL per process, loaded in torch's order.  Called when the C ABI's collective is first used (nwknn_comm_create dlopens
    "librccl.so.1", the copy already in the process first).  PyTorch-ROCm bundles its own librccl.so under that SONAME.  Two orders end in
    `free(): invalid pointer` / `double free or corruption` (SIGABRT) in the libraries' static destructors at process exit -- AFTER every
    handle has been destroyed cleanly, so nothing a destroy call does can prevent it:
      * /opt/rocm's RCCL loaded first, `import torch` afterwards (the system RCCL next to torch's copies of its dependencies:
        profiles/r04d_exit_fault_child_two_rccl_copies.log);
      * torch's OWN librccl.so loaded by path first, `import torch` afterwards (profiles/r04e_exit_fault_probe_import_order.log, variant D;
        variants B -- no torch import at all -- and C -- torch imported first -- exit with 0).
    What works is torch's own load order, so that is what is forced here: if torch is installed it is imported BEFORE RCCL is touched (its
    libtorch_hip brings librccl.so in as a dependency) and nwknn_comm_create then finds that copy.  Without torch the system copy is used."""
    if _torch_lib_dir() is None:
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load() -> ctypes.CDLL:
    """Load libnwknn.so once and declare every prototype of include/nwknn.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NwknnError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Build it with `python -m neighborhoodwatch_amd.build` (needs hipcc); there is no CPU fallback.")
    _bind_to_torch_rocm_runtime()
    L = ctypes.CDLL(str(LIB_PATH))
    L.nwknn_version.restype = c_int
    L.nwknn_last_error.restype = c_char_p
    L.nwknn_device_count.restype = c_int
    L.nwknn_mem_info.argtypes = [c_int, POINTER(c_size_t), POINTER(c_size_t)]
    L.nwknn_create.argtypes = [POINTER(c_void_p), c_int]
    L.nwknn_destroy.argtypes = [c_void_p]
    L.nwknn_set_stream.argtypes = [c_void_p, c_void_p]
    L.nwknn_set_option.argtypes = [c_void_p, c_char_p, c_int64]
    L.nwknn_get_stats.argtypes = [c_void_p, POINTER(NwknnStats)]
    L.nwknn_synchronize.argtypes = [c_void_p]
    L.nwknn_tile_kernel_name.argtypes = [c_void_p]
    L.nwknn_set_base.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int, c_int]
    L.nwknn_search.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int32, c_int, c_int64, c_void_p, c_void_p, c_int]
    L.nwknn_merge.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int]
    L.nwknn_set_base_columns.argtypes = [c_void_p, POINTER(c_void_p), c_int64, c_int32, c_int]
    L.nwknn_search_columns.argtypes = [c_void_p, POINTER(c_void_p), c_int64, c_int, c_int32, c_int, c_int64, c_void_p, c_void_p, c_int]
    L.nwknn_set_base_column_chunks.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int32, c_int64, c_int32, c_int]
    L.nwknn_search_column_chunks.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int32, c_int64, c_int, c_int32, c_int, c_int64,
                                             c_void_p, c_void_p, c_int]
    L.nwknn_base_begin.argtypes = [c_void_p, c_int64, c_int32, c_int]
    L.nwknn_base_append.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int]
    L.nwknn_base_append_column_chunks.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int32, c_int64, c_int]
    L.nwknn_base_end.argtypes = [c_void_p]
    L.nwknn_search_attach.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int32, c_int, c_int64, POINTER(c_int)]
    L.nwknn_search_attach_column_chunks.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int32, c_int64, c_int, c_int32, c_int, c_int64,
                                                    POINTER(c_int)]
    L.nwknn_search_collect.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
    L.nwknn_maxsim_set_docs.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_int]
    L.nwknn_maxsim_search.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int, c_int, c_int32, c_int64, c_void_p, c_void_p, c_int]
    L.nwknn_generate.argtypes = [c_void_p, c_uint64, c_int64, c_int64, c_int32, c_int, c_void_p]
    L.nwknn_comm_create.argtypes = [POINTER(c_void_p), POINTER(c_int), c_int]
    L.nwknn_comm_destroy.argtypes = [c_void_p]
    L.nwknn_comm_device_count.argtypes = [c_void_p]
    L.nwknn_comm_engine.argtypes = [c_void_p, c_int]
    L.nwknn_comm_set_option.argtypes = [c_void_p, c_char_p, c_int64]
    L.nwknn_comm_set_base.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int]
    L.nwknn_comm_generate_base.argtypes = [c_void_p, c_uint64, c_int64, c_int32, c_int]
    L.nwknn_search_sharded.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int32, c_int, c_void_p, c_void_p]
    L.nwknn_comm_get_stats.argtypes = [c_void_p, POINTER(NwknnCommStats)]
    L.nwknn_pair_distances.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int32, c_int, c_int64, c_void_p, c_int]
    L.nwknn_pair_dots.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int32, c_int64, c_void_p, c_int]
    L.nwknn_mfma_ceiling.argtypes = [c_void_p, c_int, c_double, POINTER(c_double), POINTER(c_double)]
    L.nwknn_fabric_probe.argtypes = [c_void_p, c_int64, c_double, POINTER(c_double)]
    L.nwknn_comm_merge_lists.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int, c_void_p, c_void_p]
    L.nwknn_comm_maxsim_set_docs.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int]
    L.nwknn_maxsim_search_sharded.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int, c_int32, c_void_p, c_void_p]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("nwknn_last_error", "nwknn_tile_kernel_name", "nwknn_comm_engine"):
            fn.restype = c_int
    L.nwknn_comm_engine.restype = c_void_p
    L.nwknn_tile_kernel_name.restype = c_char_p
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    """Map a negative NWKNN_E* code to a Python exception (ValueError for bad arguments, like the
    reference's assertions at cu_knn.py:136,177,237; RuntimeError otherwise)."""
    if rc == NWKNN_OK:
        return
    msg = load().nwknn_last_error().decode("utf-8", "replace")
    text = f"{what} failed (code {rc}): {msg}"
    if rc == -1:
        raise ValueError(text)
    if rc == -3:
        raise MemoryError(text)
    raise NwknnError(text)
