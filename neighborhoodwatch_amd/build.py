"""Builds libnwknn.so in-tree with hipcc for gfx950 (the only target; no other arch, no CPU build).

    python -m neighborhoodwatch_amd.build [--force]
    python -m neighborhoodwatch_amd.build --dev      # ab/libnwknn_dev.so with -DNWKNN_DEV (timing ablations, in-kernel stamps: tools/ only)

Every source file is compiled to its own object (in parallel, only when it or a header is newer than the object) and the
objects are linked into the shared library, so touching one kernel file rebuilds one object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
OUT = PKG / "libnwknn.so"
SOURCES = [CSRC / n for n in ("nwk_tile_f16x_pre.hip", "nwk_tile_f16x_raw.hip", "nwk_tile_f16y_pre.hip", "nwk_tile_f16y_raw.hip", "nwk_tile_f16.hip", "nwk_tile_f16p.hip", "nwk_maxsim.hip", "nwk_tile_f32.hip", "nwk_select.hip", "nwk_prep.hip", "nwk_calib.hip", "nwknn_api.cpp", "nwknn_comm.cpp")]
HEADERS = [CSRC / "nwk_device.h", CSRC / "nwk_f16_common.h", CSRC / "nwk_tile_f16x.h", CSRC / "nwk_tile_f16y.h", CSRC / "nwknn_internal.h", PKG.parent / "include" / "nwknn.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=off", "-x", "hip"]


def hipcc_path() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _obj(src: Path) -> Path:
    return OBJ / (src.stem + ".o")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(p.stat().st_mtime > t for p in deps)


def needs_build() -> bool:
    return _stale(OUT, SOURCES + HEADERS + [Path(__file__)])


def build_dev(verbose: bool = False) -> Path:
    """Developer build: the same sources with -DNWKNN_DEV, which adds the switches whose results are wrong or unguaranteed by
    construction (options diag, dbg_ptr, pre_eps_pct < 100, prescreen_prune).  Never the product: it goes to ab/ (git-ignored), the
    package only ever loads neighborhoodwatch_amd/libnwknn.so unless NWKNN_LIB_PATH points elsewhere (tools/ do that)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out_dir = PKG.parent / "ab"
    obj_dir = out_dir / "dev_obj"
    obj_dir.mkdir(parents=True, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = obj_dir / (src.stem + ".o")
        cmd = [hipcc, *FLAGS[:-2], "-DNWKNN_DEV", "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(str(obj))
    out = out_dir / "libnwknn_dev.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out)], check=True)
    return out


def build_dev_comm(verbose: bool = False) -> Path:
    """ab/libnwknn_devcomm.so = the product's objects with ONE file, nwknn_comm.cpp, recompiled under -DNWKNN_DEV: the only build in which a
    collective transport that announces itself as a test stand-in (tests/fake_rccl) may place several shards on one GPU.  The virtual-device
    tests and the bench rehearsal load it through NWKNN_LIB_PATH; the shipped library refuses a device listed twice whatever is loaded."""
    # The product's objects are linked as they are.  They are only (re)built here when some are MISSING: a staleness check by modification
    # time would fire on a box whose copy of the tree scrambled the times, and relinking libnwknn.so in place under a process that has it
    # mapped (the test runner) is not something a helper should do.
    if not OUT.exists() or not all(_obj(s).exists() for s in SOURCES):
        build(force=False, verbose=verbose)
    hipcc = hipcc_path()
    out_dir = PKG.parent / "ab"
    out_dir.mkdir(exist_ok=True)
    out = out_dir / "libnwknn_devcomm.so"
    src = CSRC / "nwknn_comm.cpp"
    if not _stale(out, [OUT, src, *HEADERS, Path(__file__)]):
        return out
    obj = out_dir / "nwknn_comm_dev.o"
    cmd = [hipcc, *FLAGS[:-2], "-DNWKNN_DEV", "-x", "hip", "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, capture_output=True)
    objs = [str(_obj(s)) for s in SOURCES if s.name != "nwknn_comm.cpp"] + [str(obj)]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out)], check=True, capture_output=True)
    return out


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    OBJ.mkdir(exist_ok=True)

    def compile_one(src: Path):
        obj = _obj(src)
        if not force and not _stale(obj, [src, *HEADERS, Path(__file__)]):
            return None
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        return subprocess.run(cmd, capture_output=True, text=True), src

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        for res in pool.map(compile_one, SOURCES):
            if res is not None and res[0].returncode != 0:
                raise RuntimeError(f"hipcc failed on {res[1].name} ({res[0].returncode}):\n{res[0].stdout}\n{res[0].stderr}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(_obj(s)) for s in SOURCES], "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build_dev(verbose=True) if "--dev" in sys.argv else build_dev_comm(verbose=True) if "--dev-comm" in sys.argv else build(force="--force" in sys.argv, verbose=True))
