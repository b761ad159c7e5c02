"""dev: both tile shapes must give the oracle's bits."""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import cpu_oracle as o
from neighborhoodwatch_amd import KnnEngine
shapes = [(300, 5000, 768, 100), (129, 4097, 64, 7), (100, 1000, 384, 10), (64, 9000, 32, 20), (700, 3000, 100, 16), (257, 130, 1, 130)]
ok = True
for tr in (128, 256):
    with KnnEngine(0) as eng:
        eng.set_option("tile_rows", tr)
        for (q, b, d, k) in shapes:
            base, qs = o.gen(0x51 + d, 0, b, d), o.gen(0x52 + d, 0, q, d)
            eng.set_base(base)
            for metric in ("sqeuclidean", "cosine", "one_minus_dot"):
                dd, ii = eng.search(qs, k, metric)
                rd, ri = o.knn_canonical(base, qs, k, metric)
                good = np.array_equal(ii, ri) and np.array_equal(dd.view(np.uint32), rd.view(np.uint32))
                ok &= good
                if not good:
                    print("MISMATCH tile_rows", tr, (q, b, d, k), metric, (ii != ri).mean())
    print("tile_rows", tr, "done", flush=True)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
