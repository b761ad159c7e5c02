"""MaxSim late interaction (BASELINE config 5).  The reference has no MaxSim (its `ck` is flat token kNN), so there is no
reference parity to claim: the HIP kernel is checked bit for bit against the oracle's restatement of the engine's canonical
arithmetic, and the oracle against a float64 evaluation of the textbook formula."""
import numpy as np
import pytest


def _tokens(oracle, seed, n, t, d):
    return oracle.gen(seed, 0, n * t, d).reshape(n, t, d)


def test_oracle_maxsim_matches_float64_formula(oracle):
    docs, qs = _tokens(oracle, 0xC1, 300, 64, 128), _tokens(oracle, 0xC2, 7, 32, 128)
    s, i = oracle.maxsim_canonical(docs, qs, 20)
    es, ei = oracle.maxsim_exact_fp64(docs, qs, 20)
    assert np.abs(s - es).max() <= 1e-5 * np.abs(es).max()
    assert (i == ei).mean() > 0.98 and (np.diff(s, axis=1) <= 0).all()
    for q in range(len(qs)):                                   # every reported doc's fp64 score is within tolerance of the true rank's
        true = (qs[q].astype(np.float64) @ docs[i[q]].astype(np.float64).reshape(-1, 128).T).reshape(32, 20, 64).max(2).sum(0)
        assert np.abs(true - es[q]).max() <= 2e-5 * np.abs(es).max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 7, 20), (1030, 33, 100), (5000, 9, 1000), (4100, 130, 10)])
def test_hip_maxsim_matches_oracle_bit_exact(oracle, shape):
    from neighborhoodwatch_amd import KnnEngine
    nd, nq, k = shape
    docs, qs = _tokens(oracle, 0xC3 + nd, nd, 64, 128), _tokens(oracle, 0xC4 + nq, nq, 32, 128)
    with KnnEngine(0) as eng:
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(qs, k)
    rs, ri = oracle.maxsim_canonical(docs, qs, k)
    assert np.array_equal(i, ri)
    assert np.array_equal(s.view(np.uint32), rs.view(np.uint32))


@pytest.mark.gpu
def test_hip_maxsim_overflow_repair_and_offsets(oracle):
    """Docs arrive in increasing-score order so every doc beats the running threshold: lists overflow and are repaired."""
    from neighborhoodwatch_amd import KnnEngine
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 32, 128)).astype(np.float32)
    base = rng.standard_normal((2500, 64, 128)).astype(np.float32)
    docs = base * np.linspace(0.05, 3.0, 2500, dtype=np.float32)[:, None, None]     # later docs score higher for every query
    with KnnEngine(0) as eng:
        eng.set_option("capacity", 256)
        eng.set_option("first_chunk", 64)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(q, 20, doc_offset=7_000_000_000)
        st = eng.stats()
        with pytest.raises(ValueError):
            eng.maxsim_search(q, 2501)
        with pytest.raises(ValueError):
            eng.maxsim_search(q[:, :16], 5)                   # 16 tokens per query: unsupported
    rs, ri = oracle.maxsim_canonical(docs, q, 20)
    assert np.array_equal(i, ri + 7_000_000_000) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
    assert st["overflowed_queries"] > 0 and st["rescan_launches"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 7, 20), (1030, 33, 100), (5000, 9, 1000), (4100, 130, 10), (20000, 64, 50)])
def test_hip_maxsim_prescreened_matches_oracle_bit_exact(oracle, shape):
    """Option prescreen before maxsim_set_docs: fp16 MaxSim tile kernel lists (query, doc) pairs under the additive score bound
    R(query) + C(doc); every survivor is recomputed with the canonical fp32 arithmetic.  Same bits as the fp32 kernel / the oracle."""
    from neighborhoodwatch_amd import KnnEngine
    nd, nq, k = shape
    docs, qs = _tokens(oracle, 0xC3 + nd, nd, 64, 128), _tokens(oracle, 0xC4 + nq, nq, 32, 128)
    with KnnEngine(0, profile=True) as eng:
        eng.set_prescreen(True)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(qs, k)
        st = eng.stats()
    rs, ri = oracle.maxsim_canonical(docs, qs, k)
    assert st["prescreen"] == 1 and (st["rescore_ms"] > 0.0 or nd <= 512)
    assert np.array_equal(i, ri), f"doc ids differ at {(i != ri).sum()} of {i.size} positions"
    assert np.array_equal(s.view(np.uint32), rs.view(np.uint32))


@pytest.mark.gpu
def test_hip_maxsim_prescreened_overflow_repair_unit_tokens_and_offsets(oracle):
    from neighborhoodwatch_amd import KnnEngine
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 32, 128)).astype(np.float32)
    base = rng.standard_normal((2500, 64, 128)).astype(np.float32)
    docs = base * np.linspace(0.05, 3.0, 2500, dtype=np.float32)[:, None, None]     # later docs score higher for every query
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.set_option("capacity", 256)
        eng.set_option("first_chunk", 64)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(q, 20, doc_offset=7_000_000_000)
        st = eng.stats()
    rs, ri = oracle.maxsim_canonical(docs, q, 20)
    assert st["prescreen"] == 1 and st["overflowed_queries"] > 0 and st["rescan_launches"] > 0
    assert np.array_equal(i, ri + 7_000_000_000) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
    # ColBERT-like unit-norm tokens, near-duplicate documents (scores closer than the fp16 bound can separate)
    d2 = rng.standard_normal((3000, 64, 128)).astype(np.float32)
    d2[1500:] = d2[:1500] + 1e-4 * rng.standard_normal((1500, 64, 128)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=2, keepdims=True)
    q2 = rng.standard_normal((40, 32, 128)).astype(np.float32)
    q2 /= np.linalg.norm(q2, axis=2, keepdims=True)
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.maxsim_set_docs(d2)
        s, i = eng.maxsim_search(q2, 100)
        assert eng.stats()["prescreen"] == 1
    rs, ri = oracle.maxsim_canonical(d2, q2, 100)
    assert np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("waves,form", [(8, 3), (8, 2), (8, 1), (4, 1)])
@pytest.mark.parametrize("strip", [0, 3, 5, 64, 1])
def test_hip_maxsim_prescreen_kernel_forms_give_the_same_bits(oracle, strip, waves, form):
    """The forms of the d <= 128 pre-screen tile kernel (the ring kernel with auto / odd / longest strips, as the 256 x 256 eight-wave tile of
    8 queries -- round 5: pipelined (the default), deep ring, plain ring -- or the 256 x 128 four-wave tile of 4; ms_strip = 1: the any-d kernel) list under the same bound, so
    each of them must end in the oracle's bits; d = 64 is two k-slabs per tile in the four-stage ring (stage != k-slab), with a document
    count whose last strip is ragged; 21 and 6 queries leave the last query tile of either shape partly empty, 9 fills one 8-query tile
    and one query of the next."""
    from neighborhoodwatch_amd import KnnEngine
    for nd, nq, k, d in ((9000, 21, 64, 128), (3333, 6, 100, 64), (2100, 9, 30, 96)):
        docs, qs = _tokens(oracle, 0xD1 + nd, nd, 64, d), _tokens(oracle, 0xD2 + nq, nq, 32, d)
        with KnnEngine(0, profile=True) as eng:
            eng.set_prescreen(True)
            eng.set_option("ms_strip", strip)
            eng.set_option("ms_waves", waves)
            eng.set_option("ms_form", form)
            eng.maxsim_set_docs(docs)
            s, i = eng.maxsim_search(qs, k)
            assert eng.stats()["prescreen"] == 1
            ring = {(8, 3): "nwk_maxsim_pre_tile8_kernel<4>" if d > 64 else "nwk_maxsim_pre_tile8_kernel<2>", (8, 2): "nwk_maxsim_pre_ring16_kernel<8,2>", (8, 1): "nwk_maxsim_pre_ring16_kernel<8,1>",
                    (4, 1): "nwk_maxsim_pre_ring16_kernel<4,1>"}[(waves, form)]
            assert eng.tile_kernel_name() == ("nwk_maxsim_pre_kernel" if strip == 1 else ring)
        rs, ri = oracle.maxsim_canonical(docs, qs, k)
        assert np.array_equal(i, ri), f"doc ids differ at {(i != ri).sum()} of {i.size} positions (strip={strip}, d={d})"
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32))


@pytest.mark.gpu
def test_hip_maxsim_many_queries_chunks_stay_launchable():
    """70 000 queries are 17 500 query groups: a chunk of document tiles x query groups must stay under HIP's 2^32 threads per
    launch (the chunk walk bounds its chunks per kernel form); direct and pre-screened results agree bit for bit."""
    import torch
    from neighborhoodwatch_amd import KnnEngine
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    docs = torch.randn((9000, 64, 128), device="cuda", generator=g)
    qs = torch.randn((70_000, 32, 128), device="cuda", generator=g)
    res = []
    for pre in (False, True):
        with KnnEngine(0, profile=True) as eng:
            eng.set_prescreen(pre)
            eng.maxsim_set_docs(docs)
            s, i = eng.maxsim_search(qs, 10)
            assert eng.stats()["prescreen"] == int(pre)
            res.append((s.clone(), i.clone()))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("prescreen", [False, True])
def test_hip_maxsim_ragged_documents_and_queries(oracle, prescreen):
    """Documents of 1..64 tokens and queries of 1..32 tokens (float16 or float32) through the fixed 64 x 32 tile shape: documents are
    padded by repeating their last token, queries by zero tokens (KnnEngine.pad_tokens).  Bits equal the oracle on the padded
    matrices, and the scores equal the float64 MaxSim of the UNPADDED data within fp32 tolerance (padding changes nothing)."""
    from neighborhoodwatch_amd import KnnEngine
    rng = np.random.default_rng(17)
    d = 96
    docs = [rng.standard_normal((int(t), d)).astype(np.float16 if n % 3 == 0 else np.float32) for n, t in enumerate(rng.integers(1, 65, 700))]
    qs = [rng.standard_normal((int(t), d)).astype(np.float32) for t in rng.integers(1, 33, 19)]
    with KnnEngine(0) as eng:
        eng.set_prescreen(prescreen)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(qs, 25)
    pd_, pq_ = KnnEngine.pad_tokens(docs, 64, True), KnnEngine.pad_tokens(qs, 32, False)
    rs, ri = oracle.maxsim_canonical(pd_, pq_, 25)
    assert np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
    for q in (0, 7, 18):                                           # float64 MaxSim of the ragged originals
        exact = np.array([(qs[q].astype(np.float64) @ x.astype(np.float64).T).max(axis=1).sum() for x in docs])
        assert np.allclose(s[q], exact[i[q]], rtol=1e-5, atol=1e-4)
        assert np.allclose(np.sort(exact)[::-1][:25], s[q], rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError, match="tokens"):
        KnnEngine.pad_tokens([np.zeros((65, d), np.float32)], 64, True)


@pytest.mark.gpu
@pytest.mark.parametrize("prescreen", [False, True])
def test_maxsim_through_the_c_abi_collective(oracle, prescreen):
    """nwknn_comm_maxsim_set_docs / nwknn_maxsim_search_sharded: documents sharded over the communicator's devices, one all-gather per
    array, sliced merge by (score descending, doc id ascending).  With the one device of this box it must give nwknn_maxsim_search's
    bits -- RCCL, the all-gather and the descending merge all run; more devices are covered list-wise by the next test."""
    from neighborhoodwatch_amd import ShardedKnnEngine
    docs, qs = _tokens(oracle, 0xC7, 2500, 64, 128), _tokens(oracle, 0xC8, 41, 32, 128)
    rs, ri = oracle.maxsim_canonical(docs, qs, 100)
    with ShardedKnnEngine([0]) as eng:
        eng.set_prescreen(prescreen)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(qs, 100)
        st = eng.stats()
        with pytest.raises(Exception):
            eng.search(qs.reshape(-1, 128), 5)                    # the devices hold MaxSim documents, not a kNN base
    assert np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
    assert st["n_devices"] == 1 and st["allgather_ms"] > 0.0 and st["merge_ms"] > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n_lists,k,n_slices", [(8, 1000, 3), (4, 100, 2)])
def test_descending_merge_of_score_lists(n_lists, k, n_slices):
    """The merge step of a multi-device MaxSim search on caller-built lists: scores ranked high to low, equal scores by the lower doc id,
    empty slots (-inf, -1) last; sliced and (8 x 1000 entries) two-level."""
    from neighborhoodwatch_amd import ShardedKnnEngine
    rng = np.random.default_rng(n_lists + k)
    nq = 29
    pool = rng.standard_normal(300).astype(np.float32) * 50
    scores = -np.sort(-rng.choice(pool, size=(n_lists, nq, k)), axis=2)
    docs = np.empty((n_lists, nq, k), dtype=np.int64)
    for l in range(n_lists):
        for q in range(nq):
            ids = rng.choice(1_000_000, size=k, replace=False) + l * 1_000_000
            order = np.lexsort((ids, -scores[l, q]))
            docs[l, q], scores[l, q] = ids[order], scores[l, q][order]
    docs[0, :, k - 7:] = -1
    scores[0, :, k - 7:] = -np.inf
    flat_s, flat_d = scores.transpose(1, 0, 2).reshape(nq, -1), docs.transpose(1, 0, 2).reshape(nq, -1)
    es, ed = np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)
    for q in range(nq):
        ok = flat_d[q] >= 0
        order = np.lexsort((flat_d[q][ok], -flat_s[q][ok]))[:k]
        es[q], ed[q] = flat_s[q][ok][order], flat_d[q][ok][order]
    with ShardedKnnEngine([0]) as eng:
        s, d = eng.merge_lists(scores, docs, k, n_slices, descending=True)
    assert np.array_equal(d, ed) and np.array_equal(s.view(np.uint32), es.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.drives_fallback
def test_maxsim_invariant_is_counted_and_its_failure_falls_back_to_the_direct_kernel(oracle, capfd):
    """The MaxSim rescoring kernels hold every rescored (query, doc) pair's fp16 score estimate against its canonical score and the bound
    R(query) + C(doc) it was kept under (nwknn_stats.prescreen_*), like the kNN path; with the invariant held against a sliver of the bound
    (pre_check_ppm) the attempt is discarded and the direct fp32 MaxSim kernel gives the same bits."""
    from neighborhoodwatch_amd import KnnEngine
    docs, qs = _tokens(oracle, 0xD1, 6000, 64, 128), _tokens(oracle, 0xD2, 37, 32, 128)
    rs, ri = oracle.maxsim_canonical(docs, qs, 50)
    with KnnEngine(0) as eng:
        eng.set_prescreen(True)
        eng.maxsim_set_docs(docs)
        s, i = eng.maxsim_search(qs, 50)
        st = eng.stats()
        assert st["prescreen"] == 1 and st["prescreen_checked"] > 37 * 50 and st["prescreen_violations"] == 0 and st["prescreen_fallbacks"] == 0
        assert 0.0 < st["prescreen_worst"] < 0.9, st["prescreen_worst"]
        assert np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
        eng.set_option("pre_check_ppm", max(1, int(st["prescreen_worst"] * 1e6 / 4)))
        s, i = eng.maxsim_search(qs, 50)
        st = eng.stats()
        assert st["prescreen"] == 0 and st["prescreen_fallbacks"] == 1 and st["prescreen_violations"] > 0, st
        assert np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
        assert "MaxSim pre-screen invariant failed" in capfd.readouterr().err
