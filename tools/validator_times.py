"""validate_files_v0 (parquet_to_format.py:351-383) at C2 size -- 10k queries x 1M base rows, d = 768, k = 100, unit rows -- with its Q x k gather + dot
on the host (float64 numpy, the reference's arithmetic vectorised per query) and on the GPU (nwknn_pair_dots, float64; the host decides pairs within 1e-10
of the tolerance edge): same mismatch count, wall time of each.  usage: validator_times.py [base rows] [queries]"""
import sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from neighborhoodwatch_amd import KnnEngine, parquet_to_format as ptf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
d, k = 768, 100
with KnnEngine(0) as eng:
    base = eng.generate(0x6E770001, 0, B, d); queries = eng.generate(0x6E770002, 0, Q, d)
    base = base / base.norm(dim=1, keepdim=True); queries = queries / queries.norm(dim=1, keepdim=True)      # unit rows: the validator's identity holds
    eng.set_base(base)
    dist, idx = eng.search(queries, k)
    hb, hq, hd, hi = base.cpu().numpy(), queries.cpu().numpy(), dist.cpu().numpy(), idx.cpu().numpy()
hd[7, 3] += 0.01; hi[11, 5] = (hi[11, 5] + 12345) % B                  # two planted mismatches
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    for name, arr, t in (("q.fvec", hq, "f"), ("b.fvec", hb, "f"), ("i.ivec", hi.astype(np.int32), "i"), ("d.fvec", hd, "f")):
        ptf.write_ivec_fvec_from_dataframe(tmp, "m", f"{tmp}/{name}", arr, t, arr.shape[1])
    files = [f"{tmp}/{n}" for n in ("q.fvec", "b.fvec", "i.ivec", "d.fvec")]
    import contextlib, io
    res = {}
    for label, dev in (("host", -1), ("gpu", 0)):
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            n = ptf.validate_files_v0(tmp, *files, device=dev)
        res[label] = (n, time.perf_counter() - t0)
        print(f"{label}: {Q} x {k} pairs against {B} base rows: {n} mismatches in {res[label][1]:.2f} s (reading the four files included)", flush=True)
    assert res["host"][0] == res["gpu"][0] == 2
