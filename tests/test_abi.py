"""CPU: the C-ABI library loads and exports exactly what include/nwknn.h declares; the product fails loudly
without a GPU and never reaches for the oracle."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "nwknn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nwknn_[a-z_]+)\s*\(", text)))


def test_header_and_library_agree():
    from neighborhoodwatch_amd import _lib
    L = _lib.load()
    names = declared_functions()
    assert len(names) >= 14
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/nwknn.h but not exported by libnwknn.so"
    assert sorted(_lib.EXPORTS) == names
    assert L.nwknn_version() == 2
    out = ctypes.string_at(L.nwknn_last_error())
    assert isinstance(out, bytes)


def test_stats_struct_matches_header():
    from neighborhoodwatch_amd._lib import NwknnStats
    text = (ROOT / "include" / "nwknn.h").read_text()
    body = text[text.index("typedef struct nwknn_stats {"):text.index("} nwknn_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        _, names = decl.split(None, 1)
        fields += [n.strip() for n in names.split(",")]
    assert fields == [n for n, _ in NwknnStats._fields_]


def test_comm_stats_layout_and_argument_errors(tmp_path):
    """The multi-GPU entry points: struct layout as gcc sees the header == the ctypes mirror; bad arguments are refused before any
    device or RCCL is touched."""
    import subprocess
    from neighborhoodwatch_amd import _lib
    from neighborhoodwatch_amd._lib import NwknnCommStats
    fields = [name for name, _ in NwknnCommStats._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nwknn.h"\nint main(void){printf("%zu", sizeof(nwknn_comm_stats));'
                   + "".join(f'printf(" %zu", offsetof(nwknn_comm_stats, {f}));' for f in fields) + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(NwknnCommStats) and out[1:] == [getattr(NwknnCommStats, f).offset for f in fields]
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.nwknn_comm_create(ctypes.byref(h), None, 1) == -1 and b"null" in L.nwknn_last_error()
    two = (ctypes.c_int * 2)(0, 0)
    assert L.nwknn_comm_create(ctypes.byref(h), two, 2) == -1 and b"listed twice" in L.nwknn_last_error()
    assert L.nwknn_comm_create(ctypes.byref(h), two, 0) == -1
    assert L.nwknn_search_sharded(None, None, 1, 0, 1, 0, None, None) == -1
    assert L.nwknn_comm_destroy(None) == 0 and L.nwknn_comm_device_count(None) == 0
    assert L.nwknn_comm_merge_lists(None, None, None, 2, 1, 1, 1, 0, None, None) == -1 and b"null" in L.nwknn_last_error()
    assert L.nwknn_comm_maxsim_set_docs(None, None, 1, 64, 8, 0) == -1
    assert L.nwknn_maxsim_search_sharded(None, None, 1, 32, 0, 1, None, None) == -1


def test_argument_errors_do_not_need_a_gpu():
    from neighborhoodwatch_amd import _lib
    L = _lib.load()
    assert L.nwknn_create(None, 0) == -1                       # NWKNN_EINVAL, message retrievable
    assert b"null" in L.nwknn_last_error()
    assert L.nwknn_set_option(None, b"capacity", 1024) == -1
    assert L.nwknn_search(None, None, 1, 0, 0, 1, 0, 0, None, None, 0) == -1
    assert L.nwknn_pair_distances(None, None, 1, 0, 0, None, 1, 0, 0, None, 0) == -1 and b"null" in L.nwknn_last_error()
    assert L.nwknn_pair_dots(None, None, 1, 0, 0, None, 1, 0, None, 0) == -1
    import ctypes as _c
    tf = _c.c_double()
    assert L.nwknn_mfma_ceiling(None, 1, 1.0, _c.byref(tf), None) == -1
    assert L.nwknn_set_option(None, b"pre_check_ppm", 1000000) == -1 and b"null handle" in L.nwknn_last_error()    # a shipped option: gets as far as the handle check
    assert L.nwknn_destroy(None) == 0


def test_wrong_result_switches_are_not_in_the_shipped_library():
    """Timing ablations, in-kernel stamps, a shrunken pre-screen bound and unpruned rescoring exist under -DNWKNN_DEV only: the default
    build refuses them by name (no handle, hence no GPU, is needed to see that), and the build script does not define the macro."""
    from neighborhoodwatch_amd import _lib, build
    L = _lib.load()
    for name, value in ((b"diag", 1), (b"diag", 32), (b"dbg_ptr", 4096), (b"pre_eps_pct", 99), (b"prescreen_prune", 0)):
        assert L.nwknn_set_option(None, name, value) == -1
        assert b"developer builds" in L.nwknn_last_error(), (name, L.nwknn_last_error())
    assert L.nwknn_set_option(None, b"pre_eps_pct", 100) == -1 and b"null handle" in L.nwknn_last_error()   # a legal value gets as far as the handle check
    assert not any("NWKNN_DEV" in f for f in build.FLAGS)
    # ... and neither is the test seam of the collective: the marker symbol by which a stand-in transport (tests/fake_rccl) may place several
    # shards on one GPU is looked up by developer builds of nwknn_comm.cpp only (ab/libnwknn_devcomm.so), never by the product
    blob = _lib.LIB_PATH.read_bytes()
    assert b"nwknn_test_transport_shares_one_device" not in blob and b"TEST stand-in" not in blob
    assert b"NWKNN_RCCL_LIBRARY" in blob                                    # a site's own RCCL by path stays
    # a device listed twice is refused before anything is loaded, whatever NWKNN_RCCL_LIBRARY says (no GPU needed: the check comes first)
    import ctypes as _c
    h = _c.c_void_p()
    devs = (_c.c_int * 2)(0, 0)
    import os as _os
    old = _os.environ.get("NWKNN_RCCL_LIBRARY")
    _os.environ["NWKNN_RCCL_LIBRARY"] = "/nonexistent/libfake_rccl.so"
    try:
        assert L.nwknn_comm_create(_c.byref(h), devs, 2) == -1 and b"listed twice" in L.nwknn_last_error()
    finally:
        if old is None:
            _os.environ.pop("NWKNN_RCCL_LIBRARY", None)
        else:
            _os.environ["NWKNN_RCCL_LIBRARY"] = old


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from neighborhoodwatch_amd import KnnEngine, knn
    from neighborhoodwatch_amd._lib import NwknnError
    import numpy as np
    with pytest.raises(NwknnError):
        KnnEngine(0)
    with pytest.raises(NwknnError):
        knn(np.zeros((4, 4), np.float32), np.zeros((1, 4), np.float32), 1)


def test_missing_library_is_reported(monkeypatch, tmp_path):
    from neighborhoodwatch_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libnwknn.so")
    with pytest.raises(_lib.NwknnError, match="has not been built"):
        _lib.load()


def test_product_never_imports_the_oracle():
    for path in (ROOT / "neighborhoodwatch_amd").rglob("*.py"):
        src = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{path} imports the oracle"
        assert "libnwknn_oracle" not in src
    for path in (ROOT / "neighborhoodwatch_amd" / "csrc").iterdir():
        if not path.is_file():
            continue                                       # csrc/build/ holds the objects
        assert "oracle/" not in path.read_text().replace("oracle/nwknn_oracle.c", "").replace("oracle/cpu_oracle", ""), path


def test_stats_struct_layout_matches_the_header(tmp_path):
    """sizeof/offsetof of nwknn_stats as gcc sees include/nwknn.h == the ctypes mirror in _lib.py (guards against ABI drift)."""
    import ctypes
    import subprocess
    from neighborhoodwatch_amd._lib import NwknnStats
    src = tmp_path / "layout.c"
    fields = [name for name, _ in NwknnStats._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nwknn.h"\nint main(void){printf("%zu", sizeof(nwknn_stats));'
                   + "".join(f'printf(" %zu", offsetof(nwknn_stats, {f}));' for f in fields) + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(NwknnStats)
    assert out[1:] == [getattr(NwknnStats, f).offset for f in fields]
