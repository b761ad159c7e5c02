"""CPU (hipcc cross-compiles without a GPU): the register budgets the kernels' occupancy depends on, read from the compiler's own
metadata.  The fp32 tile kernel runs THREE workgroups per CU only while its 256-row shape stays within 168 VGPRs, and its speed
depended for two rounds on a handful of spilled registers whose number moved with unrelated code; the fp16 ping-pong
kernel must not touch scratch at all at its tile rate.  A toolchain bump that moves either is seen here, not on the GPU box."""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "neighborhoodwatch_amd" / "csrc"
sys.path.insert(0, str(ROOT / "tools"))


@pytest.fixture(scope="module")
def metadata():
    import isa_report
    files = [CSRC / "nwk_tile_f32.hip", CSRC / "nwk_tile_f16x_pre.hip", CSRC / "nwk_tile_f16y_pre.hip", CSRC / "nwk_tile_f16y_raw.hip"]
    with ThreadPoolExecutor(max_workers=4) as pool:
        md = list(pool.map(isa_report.kernel_metadata, files))
    return {"f32": md[0], "f16x": md[1], "f16y": {**md[2], **md[3]}}


def test_fp32_tile_kernel_keeps_three_workgroups_per_cu(metadata):
    md = metadata["f32"]
    # cosine pre-tested on the distance until round 5 -- two correctly rounded roots and a division per accumulator, 38 spilled registers;
    # then a multiplicative split (row term x column term) with the roots staged once per tile: 10; the exact pass of a flagged block taken two
    # rows at a time (two keys + two thresholds beside the division's temporaries instead of four + four): 4, the other metrics' budget.
    # What the speed depends on is asserted directly: NO scratch access inside the loop that issues the MFMAs.
    for metric, max_spill in ((0, 7), (1, 7), (2, 7)):
        k = md[f"nwk_tile_kernel<{metric}, false, 256, 16, 0, 3>"]
        assert k["vgpr"] <= 168 and k["agpr"] == 0, k          # 3 x 168 <= 512 registers per SIMD lane
        assert k["lds"] <= 160 * 1024 // 3, k                  # three 48 KiB workgroups share a CU's 160 KiB
        assert k["vgpr_spill"] <= max_spill and k["sgpr_spill"] == 0, k
        assert k["scratch_in_k_loop"] == 0, k
        assert k["mfma"] == 64 and k["lds_dma"] == 12, k       # one 16-float slab = 2 x 4 k-pairs x (4 x 2) accumulators, unrolled once; 6 LDS-DMA per slab (prologue + loop)
    for name, k in md.items():
        if name.endswith(", 0, 2>"):                           # the two-per-CU builds (dense chunk, 128-row shape) never spill
            assert k["vgpr_spill"] == 0 and k["scratch"] == 0, (name, k)
    assert md["nwk_tile_kernel<0, false, 128, 32, 0, 2>"]["lds"] == 65536


def test_fp16_ping_pong_kernel_never_touches_scratch(metadata):
    md = metadata["f16x"]
    for metric in (0, 1, 2):
        k = md[f"nwk_tile_kernel_f16x<{metric}, 1>"]
        assert k["vgpr"] <= 256 and k["vgpr_spill"] == 0 and k["sgpr_spill"] == 0 and k["scratch"] == 0 and k["scratch_in_k_loop"] == 0, k      # two waves per SIMD: 256 registers each
        assert k["lds"] <= 160 * 1024, k
        assert k["mfma"] == 256, k                              # 4 call sites x 64 MFMAs of one slab


def test_fp16_two_issuing_waves_kernel_keeps_its_k_loop_in_registers_and_in_one_block(metadata):
    """nwk_tile_kernel_f16y holds 128 accumulators and two fragment sets of 48 registers: 32 are left.  Its first build spilled LDS addresses,
    reloaded them through scratch inside the K loop -- and the compiler waits vmcnt(0) for a scratch reload, i.e. for the LDS-DMA in flight,
    five times per slab; the cause was a scalar branch around the DMA issue (it also ends the basic block the pinned instruction stream
    lives in).  Asserted here: no spill at all, and no scratch access in any loop that issues MFMAs."""
    md = metadata["f16y"]
    for metric in (0, 1, 2):
        for pre in (0, 1):
            k = md[f"nwk_tile_kernel_f16y<{metric}, {pre}>"]
            assert k["vgpr"] <= 256 and k["vgpr_spill"] == 0 and k["scratch"] == 0 and k["scratch_in_k_loop"] == 0, k
            assert k["lds"] <= 160 * 1024 and k["mfma"] == 160, k      # X (boundary), Y (first), X, Y, X (tail): 5 x 32
