import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine
Q, B, d, k = 512, 250_000, 128, 1000
eng = KnnEngine(0, profile=True)
eng.set_prescreen(True)
docs = eng.generate(1, 0, B * 64, d).view(B, 64, d); q = eng.generate(2, 0, Q * 32, d).view(Q, 32, d)
eng.maxsim_set_docs(docs)
dbg = torch.zeros(4, dtype=torch.int64, device="cuda")
eng.set_option("dbg_ptr", dbg.data_ptr())
eng.maxsim_search(q, k)
eng.set_option("dbg_ptr", 0)
s = eng.stats()
c = dbg.cpu().numpy()
print(f"listed {c[0]} ({c[0]/Q:.0f}/query), rescored {c[1]} ({c[1]/Q:.0f}/query); tile {s['tile_ms']:.1f} ms rescore {s['rescore_ms']:.1f} ms -> {s['rescore_ms']*1e3/max(c[1],1):.2f} us per rescored pair (all waves), launches {s['tile_launches']}")
