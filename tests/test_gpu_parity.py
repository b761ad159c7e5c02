"""GPU parity: HIP path vs the CPU oracle, bit for bit (indices AND distances), through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

METRICS = ["sqeuclidean", "one_minus_dot", "cosine"]


def _bitwise(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("shape", [(100, 1000, 384, 10), (37, 2000, 100, 16), (9, 130, 36, 130), (300, 5000, 768, 100),
                                   (1, 129, 1, 1), (129, 4097, 33, 7)])
def test_matches_oracle_bit_exact(engine, oracle, shape, metric):
    q, b, d, k = shape
    base = oracle.gen(0x51 + d, 0, b, d)
    queries = oracle.gen(0x52 + d, 0, q, d)
    engine.set_base(base)
    dist, idx = engine.search(queries, k, metric)
    rd, ri = oracle.knn_canonical(base, queries, k, metric)
    assert np.array_equal(idx, ri), f"indices differ at {(idx != ri).sum()} of {idx.size} positions"
    assert _bitwise(dist, rd), "distances are not bit-identical to the canonical fmaf chain"


@pytest.mark.parametrize("tile_rows", [128, 256])
@pytest.mark.parametrize("shape", [(300, 5000, 768, 100), (257, 4097, 48, 7), (2100, 3000, 100, 16), (64, 9000, 32, 20)])
def test_both_tile_shapes_match_oracle(oracle, shape, tile_rows):
    """The 128x128/BK32 and 256x128/BK16 workgroup tiles accumulate every dot product in the same ascending-k order."""
    from neighborhoodwatch_amd import KnnEngine
    q, b, d, k = shape
    base = oracle.gen(0x61 + d, 0, b, d)
    queries = oracle.gen(0x62 + d, 0, q, d)
    with KnnEngine(0) as eng:
        eng.set_option("tile_rows", tile_rows)
        eng.set_base(base)
        for metric in METRICS:
            dist, idx = eng.search(queries, k, metric)
            rd, ri = oracle.knn_canonical(base, queries, k, metric)
            assert np.array_equal(idx, ri)
            assert _bitwise(dist, rd)
