"""dev: how selective are the pre-screen epilogue's tests on a C3-like search?  Per wave-tile (128 queries x 64 base rows): does the block
mask flag anything, would a single lane-wide test (min row term of the lane + min column term) flag it, and how many 16-row blocks are
flagged.  Developer build of the library.  usage: NWKNN_LIB_PATH=ab/libnwknn_dev.so python tools/f16x_census.py [B] [Q] [d]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 32_768
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
eng = KnnEngine(0, profile=True)
base = eng.generate(0x6E770001, 0, B, d); qs = eng.generate(0x6E770002, 0, Q, d)
eng.set_prescreen(True)
eng.set_option("f16_mfma", 4)
eng.set_base(base)
dbg = torch.zeros(((1 << 23) + 64,), dtype=torch.int64, device="cuda")
eng.set_option("dbg_ptr", dbg.data_ptr())
eng.set_option("diag", 64)
eng.search(qs, 100)
torch.cuda.synchronize()
c = dbg[(1 << 23):(1 << 23) + 4].cpu().numpy()
print(f"B={B} Q={Q} d={d}: wave-tiles {c[0]}; block mask fires in {c[1] / c[0]:.3%}; lane-wide coarse test fires in {c[2] / c[0]:.3%}; "
      f"flagged 16-row blocks per wave-tile {c[3] / c[0]:.4f} (of 8)")
p4 = 1 - (1 - c[1] / c[0]) ** 4
print(f"P(at least one of a group's four waves has a flagged block) = {p4:.3%}; with the coarse test {1 - (1 - c[2] / c[0]) ** 4:.3%}")
