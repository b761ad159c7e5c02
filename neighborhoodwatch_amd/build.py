"""Builds libnwknn.so in-tree with hipcc for gfx950 (the only target; no other arch, no CPU build).

    python -m neighborhoodwatch_amd.build [--force]
"""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "libnwknn.so"
SOURCES = [CSRC / "nwknn_kernels.hip", CSRC / "nwknn_api.cpp"]
HEADERS = [CSRC / "nwknn_internal.h", PKG.parent / "include" / "nwknn.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-ffp-contract=off", "-x", "hip"]


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *FLAGS, *map(str, SOURCES), "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
