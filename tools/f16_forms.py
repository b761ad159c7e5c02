"""fp16 tile kernel forms against each other (option f16_mfma: 1 burst, 3 streaming, 4 ping-pong): pre-screened fp32 search or fp16-input
search, interleaved rounds in one process on random data; every form's result must be the same bits as the first's.
usage: f16_forms.py "<forms...>" [d] [B] [Q] [rounds] [prescreen|f16] [f16_strip]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neighborhoodwatch_amd import KnnEngine

forms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3 4").replace(",", " ").split()]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 32_768
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
mode = sys.argv[6] if len(sys.argv) > 6 else "prescreen"
strip = int(sys.argv[7]) if len(sys.argv) > 7 else 0
eng = KnnEngine(0, profile=True)
dt = "f16" if mode == "f16" else "f32"
base = eng.generate(0x6E770001, 0, B, d, dt)
queries = eng.generate(0x6E770002, 0, Q, d, dt)
if mode != "f16":
    eng.set_prescreen(True)
if strip:
    eng.set_option("f16_strip", strip)
eng.set_base(base)
ref = None
acc = {f: [] for f in forms}
for r in range(rounds + 1):
    for f in forms:
        eng.set_option("f16_mfma", f)
        dist, idx = eng.search(queries, 100)
        s = eng.stats()
        if r == 0:
            if ref is None:
                ref = (dist.clone(), idx.clone())
            else:
                same = torch.equal(ref[1], idx) and torch.equal(ref[0].view(torch.int32), dist.view(torch.int32))
                print(f"form {f}: result == form {forms[0]}: {same}", flush=True)
            continue
        acc[f].append((s["tile_ms"], s["tile_flops"], s["total_ms"], s["rescore_ms"]))
for f in forms:
    tf = [fl / 1e12 / (ms / 1e3) for ms, fl, _, _ in acc[f]]
    tot = [t for _, _, t, _ in acc[f]]
    print(f"f16_mfma={f} {eng.tile_kernel_name() if f == forms[-1] else ''} d={d} B={B} Q={Q} {mode}: tile TF " + " ".join(f"{x:.0f}" for x in tf) +
          f" | mean {sum(tf) / len(tf):.1f} | search ms " + " ".join(f"{x:.1f}" for x in tot), flush=True)
