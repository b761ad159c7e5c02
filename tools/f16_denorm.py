"""Does v_mfma_f32_16x16x32_f16 flush fp16 subnormal inputs?  dot of rows of 2^-20 (subnormal in fp16) with rows of 2^10."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from neighborhoodwatch_amd import KnnEngine
d = 64
base = np.full((300, d), 2.0 ** -20, dtype=np.float16)
base[1] = 2.0 ** -24                      # smallest subnormal
base[2] = 2.0 ** -14                      # smallest normal
q = np.full((4, d), 2.0 ** 10, dtype=np.float16)
for shape in (1, 0):
    with KnnEngine(0) as eng:
        eng.set_option("f16_mfma", shape)
        eng.set_base(base)
        dist, idx = eng.search(q, 300, "one_minus_dot")
    dot = 1.0 - dist[0].astype(np.float64)
    order = np.argsort(idx[0])
    dot = dot[order]
    print(f"f16_mfma={shape}: dot(row of 2^-20) = {dot[0]!r} (exact {d * 2.0**-10}), dot(2^-24) = {dot[1]!r} (exact {d * 2.0**-14}), dot(2^-14) = {dot[2]!r} (exact {d * 2.0**-4})")
