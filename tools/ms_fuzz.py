"""Soak of the MaxSim pre-screen kernels: random shapes, token scales and kernel options; the pre-screened result must equal the direct
fp32 kernel's bit for bit (the direct kernel is pinned to the oracle by tests/test_maxsim.py).  usage: ms_fuzz.py [cases] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from neighborhoodwatch_amd import KnnEngine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    nd = int(rng.integers(520, 40000)); nq = int(rng.integers(1, 90)); d = int(rng.choice([32, 64, 96, 128, 128, 128]))
    k = int(min(nd, rng.choice([1, 10, 100, 333, 1000])))
    strip = int(rng.choice([0, 0, 1, 2, 5, 16, 64])); valu = int(rng.integers(0, 2))          # ms_strip 1 = the any-d kernel; ms_rescore_valu: vector-pipe rescoring
    kind = int(rng.integers(0, 3))
    g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(1 << 30)))
    docs = torch.randn((nd, 64, d), device="cuda", generator=g); qs = torch.randn((nq, 32, d), device="cuda", generator=g)
    if kind == 1:                                    # unit-norm tokens (ColBERT)
        docs = docs / docs.norm(dim=2, keepdim=True); qs = qs / qs.norm(dim=2, keepdim=True)
    elif kind == 2:                                  # documents of very different magnitudes, near-duplicates
        docs = docs * torch.logspace(-2, 1, nd, device="cuda")[:, None, None]
        docs[nd // 2:nd // 2 + nd // 4] = docs[:nd // 4] * (1 + 1e-4)
    res = []
    for pre in (0, 1):
        with KnnEngine(0, profile=True) as eng:
            eng.set_prescreen(bool(pre))
            eng.set_option("ms_strip", strip); eng.set_option("ms_rescore_valu", valu); eng.set_option("ms_waves", int(rng.choice([8, 8, 4]))); eng.set_option("ms_form", int(rng.choice([3, 3, 2, 1])))
            if rng.integers(0, 4) == 0:
                eng.set_option("capacity", 4096 if k > 256 else 1024); eng.set_option("first_chunk", 256)
            eng.maxsim_set_docs(docs)
            s, i = eng.maxsim_search(qs, k)
            st = eng.stats()
            res.append((s.clone(), i.clone(), st["prescreen"], st["prescreen_violations"], st["prescreen_fallbacks"], st["prescreen_worst"]))
    ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[1][3] == 0 and res[1][4] == 0 and res[1][5] < 1.0      # + the invariant held
    if not ok:
        bad += 1
    print(f"case {c:3d}: docs {nd:6d} queries {nq:3d} d {d:3d} k {k:4d} kind {kind} strip {strip:2d} rescore_valu {valu} prescreened {res[1][2]} worst {res[1][5]:.4f} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{cases - bad} of {cases} equal", flush=True)
sys.exit(1 if bad else 0)
