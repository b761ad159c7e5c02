"""dev: per-workgroup stamps of the fp16 16x16x32 tile kernel (pre-screen or fp16 path) on its biggest launch: phase cycles,
clock, and the idle gap between consecutive workgroups on the same CU.   usage: stamps2.py [prescreen|f16] [d]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from neighborhoodwatch_amd import KnnEngine
mode = sys.argv[1] if len(sys.argv) > 1 else "prescreen"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
Q, B, k = 32_768, 1_048_576, 100
eng = KnnEngine(0)
if mode == "prescreen":
    eng.set_prescreen(True)
else:
    eng.set_compute_dtype("fp16")
base = eng.generate(1, 0, B, d); qs = eng.generate(2, 0, Q, d)
eng.set_base(base)
eng.search(qs, k)
nq_t, nb_t = Q // 256, 2048                                  # biggest launch: 524288 rows
grid = ((nq_t + 7) // 8) * ((nb_t + 7) // 8); grid = (grid + 7) // 8 * 8 * 64
dbg = torch.zeros((grid * 4, 8), dtype=torch.int64, device="cuda")
eng.set_option("dbg_ptr", dbg.data_ptr())
eng.search(qs, k)
eng.set_option("dbg_ptr", 0)
torch.cuda.synchronize()
raw = dbg.cpu().numpy()
a = raw[:grid]
per_wave = raw[grid:grid * 3].reshape(grid, 8, 2)[a[:, 7] == 1]
a = a[a[:, 7] == 1]
tot = a[:, 3] - a[:, 0]; pro = a[:, 1] - a[:, 0]; loop = a[:, 2] - a[:, 1]; epi = a[:, 3] - a[:, 2]
rt = (a[:, 5] - a[:, 4]).astype(np.float64)
ok = rt > 0
clk = tot[ok] / rt[ok] * 100e6 / 1e9
print(f"{mode} d={d}: {len(a)} workgroups; clock GHz median {np.median(clk):.3f} (p10 {np.percentile(clk,10):.3f}, p90 {np.percentile(clk,90):.3f})")
for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue(all waves)", epi), ("total", tot)):
    print(f"  {name:20s} cycles: median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
ideal = (d // 32) * 32 * 16 * 2
print(f"  ideal MFMA cycles per SIMD per tile (2 waves x {d//32*32} MFMA x 16): {ideal}")
# idle gap per CU: group by (xcc, hw_id & CU/SH/SE bits)
hw = a[:, 6]
cu_key = ((hw >> 32) << 16) | (hw & 0xFF00)                   # xcc | se/sh/cu bits [15:8]
gaps = []; busy = []
for key in np.unique(cu_key):
    m = a[cu_key == key]
    m = m[np.argsort(m[:, 4])]
    g = (m[1:, 4] - m[:-1, 5]) * 10.0                          # ns between one workgroup's last wave ending and the next one starting
    gaps.append(g); busy.append((m[:, 5] - m[:, 4]) * 10.0)
gaps = np.concatenate(gaps); busy = np.concatenate(busy)
print(f"  CUs seen: {len(np.unique(cu_key))}; workgroup residency ns: median {np.median(busy):.0f}; gap to next workgroup on the same CU ns: "
      f"median {np.median(gaps):.0f} p10 {np.percentile(gaps,10):.0f} p90 {np.percentile(gaps,90):.0f}")

st2_w = per_wave[:, :, 0]; t3_w = per_wave[:, :, 1]
print("  per wave: K-loop end relative to wave 0's (cycles), then own epilogue duration (cycles); medians")
for wv in range(8):
    print(f"    wave {wv}: loop end {np.median(st2_w[:, wv] - st2_w[:, 0]):7.0f}   epilogue {np.median(t3_w[:, wv] - st2_w[:, wv]):7.0f}   (p90 {np.percentile(t3_w[:, wv] - st2_w[:, wv], 90):7.0f})")
print(f"  first wave done at {np.median(t3_w.min(axis=1) - a[:, 2]):.0f}, last at {np.median(t3_w.max(axis=1) - a[:, 2]):.0f} cycles after wave 0's loop end")
