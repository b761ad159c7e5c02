"""dev: per-workgroup cycle stamps of the v0 tile kernel on one big filtered chunk."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from neighborhoodwatch_amd import KnnEngine
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
diag = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Q, B, d, k = (10_000, 1_000_000, 768, 100) if dtype == "f32" else (32_768, 1_048_576, 1536, 100)
variant = dtype
eng = KnnEngine(0)
eng.set_option("diag", diag)
base = eng.generate(1, 0, B, d, dtype); qs = eng.generate(2, 0, Q, d, dtype)
eng.set_base(base)
eng.search(qs, k)
nq_t, nb_t = (Q + 255) // 256, (4096 if dtype == "f32" else 2048)          # biggest launch: 524288 rows
grid = ((nq_t + 7) // 8) * ((nb_t + 7) // 8); grid = (grid + 7) // 8 * 8 * 64
dbg = torch.zeros((grid * 4, 8), dtype=torch.int64, device="cuda")
eng.set_option("dbg_ptr", dbg.data_ptr())
eng.search(qs, k)
eng.set_option("dbg_ptr", 0)
torch.cuda.synchronize()
a = dbg.cpu().numpy()[:grid]
a = a[a[:, 7] == 1]
# the last launch to write each slot wins; slots of the largest launch dominate
cyc_total = a[:, 3] - a[:, 0]; pro = a[:, 1] - a[:, 0]; loop = a[:, 2] - a[:, 1]; epi = a[:, 3] - a[:, 2]
rt = (a[:, 5] - a[:, 4]).astype(np.float64)
ok = rt > 0
clk = cyc_total[ok] / rt[ok] * 100e6 / 1e9
print(f"variant {variant}: {len(a)} workgroups; clock GHz median {np.median(clk):.3f} (p10 {np.percentile(clk,10):.3f}, p90 {np.percentile(clk,90):.3f})")
for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("total", cyc_total)):
    print(f"  {name:9s} cycles: median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
mfma = (d // 2) * 8 * 64 if dtype == "f32" else (d // 16) * 8 * 32
print(f"  ideal MFMA cycles per wave per tile: {mfma}; x2 (two waves share a SIMD): {2*mfma}")

if dtype == "f32":
    hw = a[:, 6]
    cu_key = ((hw >> 32) << 16) | (hw & 0xFF00)
    util = []; gaps = []
    for key in np.unique(cu_key):
        m = a[cu_key == key]
        span = (m[:, 5].max() - m[:, 4].min()) * 10.0
        util.append(((m[:, 5] - m[:, 4]) * 10.0).sum() / (2.0 * span))
        # two slots per CU: greedy assignment of workgroups (by start time) to the slot that frees first
        m = m[np.argsort(m[:, 4])]
        free = [None, None]
        for st, en in zip(m[:, 4], m[:, 5]):
            j = 0 if (free[0] is None or (free[1] is not None and free[0] <= free[1])) else 1
            if free[j] is not None: gaps.append((st - free[j]) * 10.0)
            free[j] = en
    gaps = np.array(gaps)
    print(f"  CUs seen {len(util)}; slot occupancy (sum of workgroup residency / 2 / span): median {np.median(util):.3f}; "
          f"gap slot-free -> next workgroup start ns: median {np.median(gaps):.0f} p10 {np.percentile(gaps,10):.0f} p90 {np.percentile(gaps,90):.0f}")
