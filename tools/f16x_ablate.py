"""Where does the ping-pong fp16 tile kernel (f16_mfma = 4) spend a slab period?  Timing ablations behind the dev option `diag` (developer
build of the library only: python -m neighborhoodwatch_amd.build --dev; results are wrong by construction): 1 no fragment reads, 2 no
LDS-DMA inside the loop, 4 no MFMAs, 8 no epilogue.  Interleaved rounds in one process, random data.
usage: NWKNN_LIB_PATH=ab/libnwknn_dev.so python tools/f16x_ablate.py [d] [B] [Q] [rounds] [form]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neighborhoodwatch_amd import KnnEngine

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 32_768
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
form = int(sys.argv[5]) if len(sys.argv) > 5 else 4
eng = KnnEngine(0, profile=True)
base = eng.generate(0x6E770001, 0, B, d)
queries = eng.generate(0x6E770002, 0, Q, d)
eng.set_prescreen(True)
eng.set_option("f16_mfma", form)
eng.set_base(base)
variants = [(0, "nothing removed"), (8, "epilogue"), (9, "epilogue + fragment reads"), (10, "epilogue + LDS-DMA"), (12, "epilogue + MFMAs"),
            (11, "epilogue + reads + DMA (MFMAs and barriers left)"), (14, "epilogue + DMA + MFMAs (reads and barriers left)"),
            (13, "epilogue + reads + MFMAs (DMA and barriers left)"), (15, "all (barriers left)")]
acc = {v: [] for v, _ in variants}
for r in range(rounds + 1):
    for v, _ in variants:
        eng.set_option("diag", v)
        eng.search(queries, 100)
        s = eng.stats()
        if r:
            acc[v].append(s["tile_flops"] / 1e12 / (s["tile_ms"] / 1e3))
base_tf = sum(acc[0]) / len(acc[0])
for v, name in variants:
    tf = sum(acc[v]) / len(acc[v])
    print(f"diag={v:2d} removed: {name:55s} {tf:7.0f} TF-equivalent  ({(tf / base_tf - 1) * 100:+.1f} %)", flush=True)
