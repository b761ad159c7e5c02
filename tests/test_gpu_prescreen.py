"""Option "prescreen": fp16 MFMA pre-screen + canonical fp32 rescoring.  The contract is the one of the direct fp32 kernel:
indices AND distances bit-identical to the CPU oracle's canonical fmaf chain -- the fp16 pass may only ever list MORE pairs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ADDITIVE = ["sqeuclidean", "one_minus_dot", "cosine"]       # all three metrics are pre-screened


def _bitwise(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.fixture()
def pre_engine():
    from neighborhoodwatch_amd import KnnEngine
    with KnnEngine(0) as eng:
        eng.set_option("prescreen", 1)
        eng.set_option("profile", 1)
        yield eng


def _check(eng, oracle, base, queries, k, metric, expect_pre=True):
    dist, idx = eng.search(queries, k, metric)
    rd, ri = oracle.knn_canonical(base, queries, k, metric)
    assert eng.stats()["prescreen"] == (1 if expect_pre else 0)
    assert np.array_equal(idx, ri), f"{metric}: indices differ at {(idx != ri).sum()} of {idx.size} positions"
    assert _bitwise(dist, rd), f"{metric}: distances are not bit-identical"


@pytest.mark.parametrize("metric", ADDITIVE)
@pytest.mark.parametrize("shape", [(100, 1000, 384, 10), (37, 2000, 100, 16), (9, 130, 36, 130), (300, 5000, 768, 100),
                                   (1, 129, 1, 1), (129, 4097, 33, 7), (64, 20000, 64, 100), (520, 9000, 1024, 50)])
def test_prescreen_matches_oracle_bit_exact(pre_engine, oracle, shape, metric):
    q, b, d, k = shape
    base = oracle.gen(0x71 + d, 0, b, d)
    queries = oracle.gen(0x72 + d, 0, q, d)
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, k, metric)


def test_prescreen_actually_runs_the_fp16_kernel_and_rescoring(pre_engine, oracle):
    base = oracle.gen(1, 0, 30000, 128)
    queries = oracle.gen(2, 0, 300, 128)
    pre_engine.set_base(base)
    pre_engine.search(queries, 10, "sqeuclidean")
    st = pre_engine.stats()
    assert st["prescreen"] == 1 and st["rescore_ms"] > 0.0 and st["tile_launches"] >= 3


def test_cosine_with_zero_and_degenerate_rows(pre_engine, oracle):
    """Zero rows give 0/0 = NaN (ranked last), rows with a squared norm outside [2^-80, 2^80] give whatever the canonical fp32
    arithmetic gives (inf, NaN, huge): the pre-screen lists every pair involving such a row and lets the rescoring decide."""
    base = oracle.gen(3, 0, 3000, 96).copy()
    queries = oracle.gen(4, 0, 50, 96).copy()
    base[5] = 0.0
    base[77] *= 1e-30                       # squared norm underflows to 0 / denormal in fp32
    base[200] *= 1e-15
    base[1234] *= 5e11                      # squared norm above 2^80 (the matrix maximum stays below 2^40)
    queries[3] = 0.0
    queries[7] *= 1e-25
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 12, "cosine")
    pre_engine.set_base(base[:40])
    _check(pre_engine, oracle, base[:40], queries, 40, "cosine")    # k = B: the NaN / degenerate rows must all come back, in order


@pytest.mark.parametrize("metric", ADDITIVE)
def test_unit_norm_rows(pre_engine, oracle, metric):
    """The reference's inputs are unit-norm embeddings (tests/test_02_cu_knn.py:38-41): the regime the bound is tightest in."""
    base = oracle.gen_unit_rows(0xA1, 0, 20000, 384)
    queries = oracle.gen_unit_rows(0xA2, 0, 200, 384)
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 100, metric)


@pytest.mark.parametrize("metric", ADDITIVE)
def test_near_duplicates_defeat_the_fp16_pass_but_not_the_result(pre_engine, oracle, metric):
    """Rows that differ far below fp16 resolution: the pre-screen cannot separate them and lists (almost) everything, lists
    overflow and get repaired; the result must still be the canonical fp32 one."""
    rng = np.random.default_rng(5)
    centre = rng.standard_normal(64).astype(np.float32)
    base = (centre[None, :] + 1e-5 * rng.standard_normal((6000, 64))).astype(np.float32)
    queries = (centre[None, :] + 1e-5 * rng.standard_normal((40, 64))).astype(np.float32)
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 20, metric)
    assert pre_engine.stats()["overflowed_queries"] > 0


@pytest.mark.parametrize("metric", ADDITIVE)
def test_exact_ties_and_duplicated_rows(pre_engine, oracle, metric):
    base = oracle.gen(0xB1, 0, 2000, 48)
    base = np.concatenate([base, base[:700], base[100:300]])       # exact duplicates: lower index must win
    queries = np.concatenate([oracle.gen(0xB2, 0, 30, 48), base[5:25]])
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 33, metric)


@pytest.mark.parametrize("metric", ADDITIVE)
@pytest.mark.parametrize("scale_q,scale_b", [(1e-4, 1e3), (3e5, 2e-6), (1.0, 1e9)])
def test_mixed_magnitudes(pre_engine, oracle, metric, scale_q, scale_b):
    """Per-matrix power-of-two scaling keeps the fp16 copies in range whatever the magnitudes of the two sides."""
    rng = np.random.default_rng(11)
    base = (scale_b * rng.standard_normal((5000, 80))).astype(np.float32)
    base[::7] *= 1e-3                                                # rows far below the matrix maximum
    queries = (scale_q * rng.standard_normal((70, 80))).astype(np.float32)
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 25, metric)


def test_wide_dynamic_range_inside_rows(pre_engine, oracle):
    """Elements spanning 12 decades inside one row: most of them fall below the fp16 range of the scaled copy."""
    rng = np.random.default_rng(12)
    base = (rng.standard_normal((4000, 128)) * 10.0 ** rng.uniform(-9, 3, size=(4000, 128))).astype(np.float32)
    queries = (rng.standard_normal((50, 128)) * 10.0 ** rng.uniform(-9, 3, size=(50, 128))).astype(np.float32)
    pre_engine.set_base(base)
    for metric in ADDITIVE:
        _check(pre_engine, oracle, base, queries, 15, metric)


def test_non_finite_or_extreme_inputs_fall_back_to_the_direct_kernel(pre_engine, oracle):
    base = oracle.gen(0xC1, 0, 3000, 40).copy()
    queries = oracle.gen(0xC2, 0, 70, 40).copy()
    bad = base.copy()
    bad[17, 3] = np.nan
    pre_engine.set_base(bad)
    _check(pre_engine, oracle, bad, queries, 25, "sqeuclidean", expect_pre=False)
    pre_engine.set_base(base)
    badq = queries.copy()
    badq[9, :] = 1e30
    _check(pre_engine, oracle, base, badq, 25, "sqeuclidean", expect_pre=False)
    _check(pre_engine, oracle, base, queries, 25, "sqeuclidean", expect_pre=True)      # and back


def test_index_offset_columns_and_device_outputs(pre_engine, oracle):
    import torch
    base = oracle.gen(0xD1, 0, 7000, 72)
    queries = oracle.gen(0xD2, 0, 90, 72)
    rd, ri = oracle.knn_canonical(base, queries, 9, "sqeuclidean")
    pre_engine.set_base_columns([np.ascontiguousarray(base[:, j]) for j in range(base.shape[1])])
    dist, idx = pre_engine.search_columns([np.ascontiguousarray(queries[:, j]) for j in range(queries.shape[1])], 9, "sqeuclidean",
                                          index_offset=123456789012)
    assert np.array_equal(idx, ri + 123456789012) and _bitwise(dist, rd)
    tb = torch.from_numpy(base).cuda()
    tq = torch.from_numpy(queries).cuda()
    pre_engine.set_base(tb)
    dist, idx = pre_engine.search(tq, 9, "sqeuclidean")
    assert pre_engine.stats()["prescreen"] == 1
    assert np.array_equal(idx.cpu().numpy(), ri) and _bitwise(dist.cpu().numpy(), rd)


def test_small_capacity_forces_repair(oracle):
    from neighborhoodwatch_amd import KnnEngine
    base = oracle.gen(0xE1, 0, 40000, 32)
    queries = oracle.gen(0xE2, 0, 300, 32)
    with KnnEngine(0) as eng:
        eng.set_option("prescreen", 1)
        eng.set_option("capacity", 512)
        eng.set_option("first_chunk", 256)
        eng.set_base(base)
        for metric in ADDITIVE:
            dist, idx = eng.search(queries, 100, metric)
            rd, ri = oracle.knn_canonical(base, queries, 100, metric)
            assert np.array_equal(idx, ri) and _bitwise(dist, rd)


def test_k_1000(pre_engine, oracle):
    base = oracle.gen(0xF1, 0, 12000, 128)
    queries = oracle.gen(0xF2, 0, 40, 128)
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 1000, "one_minus_dot")


@pytest.mark.parametrize("shape_opt", [1, 3, 4])
def test_both_fp16_kernel_forms(oracle, shape_opt):
    """f16_mfma = 1: the burst form, 3: the streaming form (the default below 768 halves per row), 4: the ping-pong form (the default from there on), all on v_mfma_f32_16x16x32_f16 --
    pre-screened results are the same bits; the raw fp16 sums are checked against the float64 truth with a tolerance.  (Round 3 removed
    the superseded 32x32x16 and two-workgroup builds: nwknn_set_option refuses their numbers.)"""
    from neighborhoodwatch_amd import KnnEngine
    base = oracle.gen(0x91, 0, 9000, 200)
    queries = oracle.gen(0x92, 0, 300, 200)
    with KnnEngine(0) as eng:
        eng.set_option("f16_mfma", shape_opt)
        eng.set_prescreen(True)
        eng.set_base(base)
        for metric in ADDITIVE:
            dist, idx = eng.search(queries, 40, metric)
            rd, ri = oracle.knn_canonical(base, queries, 40, metric)
            assert eng.stats()["prescreen"] == 1
            assert np.array_equal(idx, ri) and _bitwise(dist, rd)
        eng.set_prescreen(False)
        eng.set_compute_dtype("fp16")
        eng.set_base(base)
        for metric in ["sqeuclidean", "one_minus_dot", "cosine"]:
            dist, idx = eng.search(queries, 40, metric)
            b16 = base.astype(np.float16).astype(np.float32); q16 = queries.astype(np.float16).astype(np.float32)
            ed = np.sort(oracle.distances_fp64(b16, q16, metric), axis=1)[:, :40]
            assert np.allclose(dist, ed, rtol=2e-3, atol=2e-3 * float(np.abs(ed).max()))


@pytest.mark.parametrize("agg", [-1, 0, 1])
@pytest.mark.parametrize("d", [72, 768])
def test_row_aggregating_build_for_loose_threshold_chunks(oracle, agg, d):
    """f16_agg: the chunks that follow the dense one (thresholds still loose: a few percent of all pairs pass) run a build of the
    streaming kernel that reserves list slots per query row, not per pair (nwknn_api.cpp run_chunk).  -1 = auto (those chunks only),
    0 = never, 1 = every filtered chunk.  Same bits whichever kernel carried a chunk -- with many queries per tile, duplicated rows
    (whole blocks pass together), a list capacity small enough to overflow, and fp16 inputs in exact mode."""
    from neighborhoodwatch_amd import KnnEngine
    base = oracle.gen(0xA61, 0, 21000, d)
    base[5000:5600] = base[17]                                   # 600 copies of one row: dense blocks of survivors far into the walk
    queries = oracle.gen(0xA62, 0, 700, d)
    queries[100:140] = base[17] + 1e-3 * queries[100:140]
    with KnnEngine(0) as eng:
        eng.set_option("f16_agg", agg)
        eng.set_option("profile", 1)
        eng.set_prescreen(True)
        eng.set_base(base)
        for metric in ADDITIVE:
            dist, idx = eng.search(queries, 50, metric)
            rd, ri = oracle.knn_canonical(base, queries, 50, metric)
            assert eng.stats()["prescreen"] == 1
            assert np.array_equal(idx, ri) and _bitwise(dist, rd), metric
        # the LAST filtered chunk of the last search (cosine = metric 2): its thresholds are loose only when forced
        expect = "nwk_tile_kernel_f16p<2,1,1>" if agg == 1 else ("nwk_tile_kernel_f16y<2,1>" if d < 768 else "nwk_tile_kernel_f16x<2,1>")
        if not os.environ.get("NWKNN_DEFAULT_OPTS"):                     # (a soak that forces a kernel form on every handle changes which one ran, not the bits)
            assert eng.tile_kernel_name() == expect
        eng.set_option("capacity", 512)                          # the 600 copies overflow a list: the repair pass re-walks with the non-aggregating build
        eng.set_option("first_chunk", 256)
        eng.set_base(base)
        dist, idx = eng.search(queries, 50, "sqeuclidean")
        rd, ri = oracle.knn_canonical(base, queries, 50, "sqeuclidean")
        assert eng.stats()["overflowed_queries"] > 0
        assert np.array_equal(idx, ri) and _bitwise(dist, rd)
    with KnnEngine(0) as eng:                                      # fp16 inputs, exact mode: canonical fp32 chain over the fp16 values
        eng.set_option("f16_agg", agg)
        eng.set_compute_dtype("fp16")
        b16 = base.astype(np.float16); q16 = queries.astype(np.float16)
        eng.set_base(b16)
        dist, idx = eng.search(q16, 50, "sqeuclidean")
        rd, ri = oracle.knn_canonical(b16.astype(np.float32), q16.astype(np.float32), 50, "sqeuclidean")
        assert np.array_equal(idx, ri) and _bitwise(dist, rd)


@pytest.mark.parametrize("d", [256, 1024, 4096])
def test_fp16_mfma_accumulation_error_is_far_inside_the_allowance(oracle, d):
    """The pre-screen bound allows the fp16 MFMA's (architecturally unspecified) internal summation an error of 4*d*u*sum|products|,
    u = 2^-24.  Measured here on the fp16 compute path (same kernel, same MFMA): 1 - dist recovers dot16; it is compared with the
    float64 dot product of the SAME fp16-rounded operands.  The epilogue's own rounding (u*|1-dot|) is charged to the error too."""
    from neighborhoodwatch_amd import KnnEngine
    rng = np.random.default_rng(d)
    base = (rng.standard_normal((1024, d)) * 8.0).astype(np.float16)
    queries = (rng.standard_normal((256, d)) * 8.0).astype(np.float16)
    with KnnEngine(0) as eng:
        eng.set_base(base)                                      # fp16 input selects the fp16 compute path
        dist, idx = eng.search(queries, 1024, "one_minus_dot")   # k = B: every pair comes back
    b64, q64 = base.astype(np.float64), queries.astype(np.float64)
    exact = q64 @ b64.T
    sum_abs = np.abs(q64) @ np.abs(b64).T
    got = 1.0 - dist.astype(np.float64)                      # dot16 of pair (q, idx[q, j])
    err = np.abs(got - np.take_along_axis(exact, idx, 1))
    unit = 2.0 ** -24 * np.take_along_axis(sum_abs, idx, 1)
    ratio = float((err / unit).max())
    assert ratio < 0.25 * 4 * d, f"fp16 MFMA accumulation error {ratio:.1f} u*sum|products| is not far inside the allowance {4 * d}"
    print(f"d={d}: max accumulation error {ratio:.2f} x u*sum|products| (allowance {4 * d})")


@pytest.mark.parametrize("metric", ADDITIVE)
def test_query_count_not_a_multiple_of_the_tile_does_not_flood_the_lists(pre_engine, oracle, metric):
    """The zero rows that pad the query matrix to 256 tie with every base row under one_minus_dot / cosine; they must not list
    anything (a first version listed every base row for them and sent the whole search through overflow repair)."""
    base = oracle.gen(0x33, 0, 30000, 64)
    queries = oracle.gen(0x34, 0, 300, 64)           # 212 padding rows
    pre_engine.set_base(base)
    _check(pre_engine, oracle, base, queries, 10, metric)
    st = pre_engine.stats()
    assert st["overflowed_queries"] == 0 and st["rescan_launches"] == 0


@pytest.mark.parametrize("metric", ADDITIVE)
def test_repair_pass_with_duplicates_in_the_lists(oracle, metric):
    """Overflow repair re-walks the base and re-finds pairs that are already listed.  Bound-based pruning must be off there: the
    k-th smallest upper bound over a list WITH duplicates can lie below the k-th distinct distance.  Tiny lists (capacity 256... the
    minimum the planner accepts) and an approaching arrival order make nearly every query overflow repeatedly."""
    from neighborhoodwatch_amd import KnnEngine
    d, B, Q, k = 24, 12000, 200, 5
    rng = np.random.default_rng(17)
    centre = rng.standard_normal(d).astype(np.float32)
    base = (centre[None, :] + rng.standard_normal((B, d)).astype(np.float32) * np.linspace(3.0, 0.02, B, dtype=np.float32)[:, None]).astype(np.float32)
    if metric == "one_minus_dot":          # "approaching" for a dot product: later rows are longer
        base = (centre[None, :] * np.linspace(0.2, 30.0, B, dtype=np.float32)[:, None]
                + 1e-4 * rng.standard_normal((B, d)).astype(np.float32)).astype(np.float32)
    queries = (centre[None, :] + 0.01 * rng.standard_normal((Q, d))).astype(np.float32)
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.set_option("capacity", 512)
        eng.set_option("first_chunk", 256)
        eng.set_base(base)
        dist, idx = eng.search(queries, k, metric)
        st = eng.stats()
        rd, ri = oracle.knn_canonical(base, queries, k, metric)
        assert st["prescreen"] == 1 and st["overflowed_queries"] > 0 and st["rescan_launches"] > 0
        assert np.array_equal(idx, ri) and _bitwise(dist, rd)


@pytest.mark.parametrize("strip", [1, 2, 3, 5])
@pytest.mark.parametrize("d", [40, 64, 192, 200])            # 1, 1, 3, 4 slabs of 64 halves: odd counts flip the LDS stage parity per tile
def test_strips_of_base_tiles_per_workgroup(oracle, strip, d):
    """Option f16_strip: a workgroup of the 16x16x32 fp16 kernel walks several consecutive base tiles, requesting the next tile's
    first slabs before the current tile's epilogue (+1 % at d=1024, +5 % at d=128; default 4)."""
    from neighborhoodwatch_amd import KnnEngine
    base = oracle.gen(0x41 + d, 0, 7000, d)                    # 28 base tiles of 256 rows: strips of 3 and 5 leave a remainder
    queries = oracle.gen(0x42 + d, 0, 300, d)
    with KnnEngine(0) as eng:
        eng.set_option("f16_strip", strip)
        eng.set_prescreen(True)
        eng.set_base(base)
        for metric in ADDITIVE:
            dist, idx = eng.search(queries, 20, metric)
            rd, ri = oracle.knn_canonical(base, queries, 20, metric)
            assert eng.stats()["prescreen"] == 1
            assert np.array_equal(idx, ri) and _bitwise(dist, rd), (metric, strip, d)


def test_operator_keyword(oracle):
    """knn(dataset, queries, k, prescreen=True): the pylibraft-shaped operator with the pre-screen switched on for one call."""
    from neighborhoodwatch_amd import knn
    from neighborhoodwatch_amd.engine import default_engine
    base = oracle.gen(0x51, 0, 5000, 96)
    queries = oracle.gen(0x52, 0, 120, 96)
    rd, ri = oracle.knn_canonical(base, queries, 15)
    d1, i1 = knn(base, queries, 15, prescreen=True)
    assert default_engine(0).stats()["prescreen"] == 1
    d2, i2 = knn(base, queries, 15)
    assert default_engine(0).stats()["prescreen"] == 0
    for d, i in ((d1, i1), (d2, i2)):
        assert np.array_equal(i, ri) and _bitwise(d, rd)


@pytest.mark.parametrize("removed", [0, 2])
def test_removed_fp16_builds_are_refused(removed):
    from neighborhoodwatch_amd import KnnEngine
    with KnnEngine(0) as eng:
        with pytest.raises(ValueError, match="f16_mfma"):
            eng.set_option("f16_mfma", removed)


def test_the_invariant_is_counted_on_every_rescored_pair(pre_engine, oracle):
    """nwknn_stats.prescreen_checked / _violations / _worst: every pair the rescoring kernel recomputes is compared with the estimate
    it was listed under.  On real hardware the fp16 MFMA's summation stays far inside the 4 d u the bound allows for it."""
    base, queries = oracle.gen(0x91, 0, 60000, 512), oracle.gen(0x92, 0, 700, 512)
    pre_engine.set_base(base)
    for metric in ADDITIVE:
        pre_engine.search(queries, 100, metric)
        st = pre_engine.stats()
        assert st["prescreen"] == 1 and st["prescreen_checked"] > 700 * 100 and st["prescreen_violations"] == 0 and st["prescreen_fallbacks"] == 0
        assert 0.0 < st["prescreen_worst"] < 0.9, (metric, st["prescreen_worst"])


@pytest.mark.parametrize("metric", ADDITIVE)
def test_deferred_rescoring_gives_the_bits_of_rescoring_every_chunk(pre_engine, oracle, metric):
    """Option "pre_defer" (default 1): only the LAST chunk of a pre-screened walk rescores; the chunks before it keep the estimates, prune on
    their bounds and filter the next chunk with the k-th smallest upper bound.  Same bits as rescoring after every chunk (and as the oracle),
    far fewer rescored pairs, no compaction between the pre-screened chunks."""
    base, queries = oracle.gen(0x93, 0, 200000, 128), oracle.gen(0x94, 0, 600, 128)
    pre_engine.set_base(base)
    rd, ri = oracle.knn_canonical(base, queries, 100, metric)
    seen = {}
    for defer in (0, 1):
        pre_engine.set_option("pre_defer", defer)
        dist, idx = pre_engine.search(queries, 100, metric)
        st = pre_engine.stats()
        assert st["prescreen"] == 1 and st["prescreen_violations"] == 0 and st["overflowed_queries"] == 0
        assert np.array_equal(idx, ri) and _bitwise(dist, rd), (metric, defer)
        seen[defer] = st
    assert seen[1]["tile_launches"] == seen[0]["tile_launches"]                          # the same walk ...
    assert 600 * 90 < seen[1]["prescreen_checked"] < 0.7 * seen[0]["prescreen_checked"]  # ... with a fraction of the rescoring


def test_deferred_rescoring_with_many_candidates_inside_the_bound(oracle):
    """Rows the fp16 estimates cannot tell apart would pile up between the chunks of a deferred walk until the lists overflow: a query whose
    pruned list grows past the pass's limit is rescored and sorted at once, as with pre_defer = 0 -- no more overflow repairs than that, and
    the canonical result either way."""
    from neighborhoodwatch_amd import KnnEngine
    rng = np.random.default_rng(23)
    centre = rng.standard_normal(96).astype(np.float32)
    base = (centre[None, :] + 3e-4 * rng.standard_normal((30000, 96))).astype(np.float32)
    base[::7] = rng.standard_normal((len(base[::7]), 96)).astype(np.float32)             # and rows far away in between
    queries = (centre[None, :] + 3e-4 * rng.standard_normal((60, 96))).astype(np.float32)
    rd, ri = oracle.knn_canonical(base, queries, 50, "sqeuclidean")
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.set_base(base)
        repaired = {}
        for defer in (1, 0):
            eng.set_option("pre_defer", defer)
            dist, idx = eng.search(queries, 50, "sqeuclidean")
            st = eng.stats()
            assert st["prescreen"] == 1
            assert np.array_equal(idx, ri) and _bitwise(dist, rd), defer
            repaired[defer] = st["overflowed_queries"]
        assert repaired[1] <= repaired[0] + 2, repaired


@pytest.mark.drives_fallback
@pytest.mark.parametrize("metric", ADDITIVE)
def test_a_failed_invariant_sends_the_search_to_the_direct_kernel_same_bits(oracle, metric, capfd):
    """Option pre_check_ppm = 10000 holds the invariant against 1 % of the bound: real pairs exceed that, the attempt is discarded, the
    search runs again on the direct fp32 kernel and says so -- and the result is the canonical one, bit for bit."""
    from neighborhoodwatch_amd import KnnEngine
    base, queries = oracle.gen(0x93, 0, 40000, 384), oracle.gen(0x94, 0, 300, 384)
    rd, ri = oracle.knn_canonical(base, queries, 50, metric)
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.set_base(base)
        eng.set_option("pre_check_ppm", 10_000)
        dist, idx = eng.search(queries, 50, metric)
        st = eng.stats()
        assert st["prescreen"] == 0 and st["prescreen_fallbacks"] == 1 and st["prescreen_violations"] > 0, st
        assert np.array_equal(idx, ri) and _bitwise(dist, rd)
        assert "pre-screen invariant failed" in capfd.readouterr().err
        eng.set_option("pre_check_ppm", 1_000_000)
        dist, idx = eng.search(queries, 50, metric)
        st = eng.stats()
        assert st["prescreen"] == 1 and st["prescreen_fallbacks"] == 0 and st["prescreen_violations"] == 0
        assert np.array_equal(idx, ri) and _bitwise(dist, rd)
        with pytest.raises(ValueError):
            eng.set_option("pre_check_ppm", 0)
