"""dev: where a slab period of the ping-pong fp16 tile kernel (f16_mfma = 4) goes, per wave group: cycle sums of the pieces of a phase
(developer build of the library: python -m neighborhoodwatch_amd.build --dev), averaged per slab over the workgroups of the biggest
launch.  usage: NWKNN_LIB_PATH=ab/libnwknn_dev.so python tools/f16x_stamps.py [d] [B] [Q]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from neighborhoodwatch_amd import KnnEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 32_768
eng = KnnEngine(0, profile=True)
base = eng.generate(0x6E770001, 0, B, d); qs = eng.generate(0x6E770002, 0, Q, d)
eng.set_prescreen(True)
eng.set_option("f16_mfma", 4)
eng.set_base(base)
eng.search(qs, 100)
grid = 1 << 20                                               # more workgroups than any launch here has (checked below)
# the biggest launch: a chunk of <= 2^20 base rows = 4096 tiles in strips of >= 1, query tiles of 256 rows, super-tiles of 64 padded to 8
assert ((Q + 255) // 256 + 15) // 16 * 16 * 4096 * 2 <= grid * 8, "debug buffer too small for this shape"
dbg = torch.zeros((grid * 8, 8), dtype=torch.int64, device="cuda")
eng.set_option("dbg_ptr", dbg.data_ptr())
eng.search(qs, 100)
s = eng.stats()
eng.set_option("dbg_ptr", 0)
torch.cuda.synchronize()
raw = dbg.cpu().numpy().reshape(grid, 8, 8)                  # [workgroup][wave][slot]; later launches overwrite earlier ones: the last (biggest) chunks remain
names = ["DMA issue", "epilogue", "fragment reads (until landed)", "wait @ LOAD barrier", "MFMA issue", "wait own DMA", "wait @ COMPUTE barrier"]
print(f"d={d} B={B} Q={Q}: tile {s['tile_flops'] / 1e12 / (s['tile_ms'] / 1e3):.0f} TF with stamps on")
for grp, waves in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    w = raw[:, waves, :].reshape(-1, 8).copy()
    tail = w[:, 7] >> 40                                         # cycles from the last stamp of the strip to the survivor lists emptied
    head = (w[:, 7] >> 16) & 0xFFFFFF                            # cycles from kernel entry to the end of the [pre] phase (slab 0 landed)
    w[:, 7] &= 0xFFFF
    tail = tail[w[:, 7] > 0]; head = head[w[:, 7] > 0]
    w = w[w[:, 7] > 0]
    per_slab = w[:, :7] / w[:, 7:8]
    tot = per_slab.sum(axis=1)
    print(f"{grp}: {len(w)} waves, slabs per strip median {np.median(w[:, 7]):.0f}; cycles per slab (median | mean):")
    for i, n in enumerate(names):
        print(f"   {n:32s} {np.median(per_slab[:, i]):8.0f} | {per_slab[:, i].mean():8.0f}")
    print(f"   {'sum':32s} {np.median(tot):8.0f} | {tot.mean():8.0f}")
    print(f"   tail of the strip (last epilogue stamp -> lists emptied), cycles per STRIP: median {np.median(tail):.0f} | mean {tail.mean():.0f}"
          f"  = {tail.mean() / (tot.mean() * np.median(w[:, 7])):.1%} of the strip's slab time")
    print(f"   head of the strip (kernel entry -> slab 0 landed, inputs staged), cycles per STRIP: median {np.median(head):.0f} | mean {head.mean():.0f}"
          f"  = {head.mean() / (tot.mean() * np.median(w[:, 7])):.1%}")
