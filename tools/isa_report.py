#!/usr/bin/env python3
"""Register / LDS budget of every kernel of a source file as the compiler reports it (amdhsa metadata of `hipcc --cuda-device-only -S`):
    python tools/isa_report.py nwk_tile_f32.hip [nwk_tile_f16x_pre.hip ...]
Needs hipcc, no GPU.  tests/test_isa_budget.py asserts the budgets the design depends on from the same metadata."""
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from neighborhoodwatch_amd import build  # noqa: E402


def scratch_in_k_loop(body: str) -> int:
    """Scratch accesses inside a LOOP that issues MFMAs: the basic blocks between a label and a backward branch to it that hold a v_mfma.
    A spilled register that is stored in the prologue and reloaded in the epilogue costs a tile a few hundred cycles; one touched in the K
    loop costs every slab."""
    lines = body.split("\n")
    label_at = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
    hits = 0
    for i, ln in enumerate(lines):
        m = re.search(r"\bs_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if not m or label_at.get(m.group(1), 10**9) > i:
            continue                                              # forward branch
        region = lines[label_at[m.group(1)]:i + 1]
        if any("v_mfma_" in x for x in region):
            hits += sum(1 for x in region if re.search(r"\bscratch_(load|store)", x))
    return hits


def kernel_metadata(src: Path, extra_flags=()):
    """{demangled-ish kernel name: {vgpr, agpr, vgpr_spill, sgpr_spill, lds, scratch, mfma}} for every kernel of one .hip file."""
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / (src.stem + ".s")
        cmd = [build.hipcc_path(), *build.FLAGS, *extra_flags, "--cuda-device-only", "-S", str(src), "-o", str(out)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr[-3000:]}")
        text = out.read_text()
    meta = text[text.index("amdhsa.kernels:"):]
    # instruction census per kernel body (between `<name>:` and its .amdhsa_kernel block)
    res = {}
    for blk in re.split(r"\n  - ", meta)[1:]:
        def field(key, default=0):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else default
        name = field("name", "")
        if not name:
            continue
        body_m = re.search(rf"^{re.escape(name)}:[^\n]*\n(.*?)^\s*\.amdhsa_kernel {re.escape(name)}\s*$", text, re.S | re.M)
        body = body_m.group(1) if body_m else ""
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        dem = re.sub(r"^void ", "", dem).split("(")[0]
        res[dem] = {"vgpr": int(field("vgpr_count")), "agpr": int(field("agpr_count")), "vgpr_spill": int(field("vgpr_spill_count")),
                    "sgpr_spill": int(field("sgpr_spill_count")), "lds": int(field("group_segment_fixed_size")),
                    "scratch": int(field("private_segment_fixed_size")), "mfma": len(re.findall(r"\bv_mfma_", body)),
                    "lds_dma": len(re.findall(r"buffer_load_dword\S* .* lds\b", body)),
                    "scratch_in_k_loop": scratch_in_k_loop(body)}
    return res


if __name__ == "__main__":
    files = [ROOT / "neighborhoodwatch_amd" / "csrc" / f for f in (sys.argv[1:] or ["nwk_tile_f32.hip"])]
    with ThreadPoolExecutor(max_workers=4) as pool:
        for src, md in zip(files, pool.map(kernel_metadata, files)):
            print(f"## {src.name}")
            for k, v in md.items():
                print(f"  {k:70s} vgpr {v['vgpr']:3d} agpr {v['agpr']:3d} spill {v['vgpr_spill']:3d} sgpr_spill {v['sgpr_spill']:3d} scratch {v['scratch']:4d} B  lds {v['lds']:6d} B  mfma {v['mfma']:4d} lds-dma {v['lds_dma']:3d} scratch ops in the K loop {v['scratch_in_k_loop']}")
