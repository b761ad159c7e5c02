"""GPU: BASELINE.json config 2 at full size (10k x 1M, d=768, k=100) checked through size-independent properties and
a sampled recomputation on the CPU oracle against the WHOLE base."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_c2_full_size_properties_and_sampled_parity(oracle):
    import torch
    from neighborhoodwatch_amd import KnnEngine
    Q, B, d, k = 10_000, 1_000_000, 768, 100
    with KnnEngine(0) as eng:
        base = eng.generate(oracle.SEED_BASE, 0, B, d)
        queries = eng.generate(oracle.SEED_QUERY, 0, Q, d)
        eng.set_base(base)
        dist, idx = eng.search(queries, k)
        dist2, idx2 = eng.search(queries, k)
        st = eng.stats()
        # queries that ARE base rows must find themselves first at distance exactly 0
        probe = base[123_456:123_456 + 64].clone()
        pd_, pi_ = eng.search(probe, 3)
        # every metric, direct kernel vs fp16 pre-screen + fp32 rescoring: the whole result must be the same bits
        direct = {m: eng.search(queries, k, m) for m in ("sqeuclidean", "one_minus_dot", "cosine")}
        eng.set_prescreen(True)
        eng.set_base(base)
        for m, (dd, di_) in direct.items():
            pd2, pi2 = eng.search(queries, k, m)
            assert eng.stats()["prescreen"] == 1 and eng.stats()["overflowed_queries"] == 0
            assert torch.equal(pi2, di_) and torch.equal(pd2.view(torch.int32), dd.view(torch.int32)), m
        eng.set_prescreen(False)
        del direct
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)                      # deterministic
    assert st["overflowed_queries"] == 0
    di, ii = dist.cpu().numpy(), idx.cpu().numpy()
    assert (np.diff(di, axis=1) >= 0).all()                                          # ascending (merge.py:104)
    assert ii.min() >= 0 and ii.max() < B
    srt = np.sort(ii, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                                         # no duplicate neighbours
    assert (pi_[:, 0].cpu().numpy() == np.arange(123_456, 123_456 + 64)).all() and (pd_[:, 0].cpu().numpy() == 0).all()
    # sampled parity: 48 query rows recomputed by the canonical CPU oracle against all 1M base rows, bit for bit
    rows = np.linspace(0, Q - 1, 48).astype(np.int64)
    hb = base.cpu().numpy()
    hq = queries[torch.from_numpy(rows).cuda()].cpu().numpy()
    rd, ri = oracle.knn_canonical(hb, hq, k)
    assert np.array_equal(ii[rows], ri)
    assert np.array_equal(di[rows].view(np.uint32), rd.view(np.uint32))
    # regenerated on the host, the inputs are what the device generator wrote
    assert np.array_equal(hb[777_000:777_004], oracle.gen_numpy(oracle.SEED_BASE, 777_000, 4, d))


@pytest.mark.timeout(1500)
def test_c3_headline_size_properties_and_sampled_parity(oracle):
    """BASELINE.json's headline config on one GPU: 100k queries x 10M base rows, d=1024, k=100 (41 GB of base in HBM).
    Size-independent properties for all rows + 32 sampled queries recomputed on the CPU against the ENTIRE base."""
    import torch
    from neighborhoodwatch_amd import KnnEngine
    Q, B, d, k = 100_000, 10_000_000, 1024, 100
    with KnnEngine(0) as eng:
        base = eng.generate(oracle.SEED_BASE, 0, B, d)
        queries = eng.generate(oracle.SEED_QUERY, 0, Q, d)
        eng.set_base(base)
        dist, idx = eng.search(queries, k)
        st = eng.stats()
        probe = base[9_876_543:9_876_543 + 16].clone()            # base rows as queries: each must find itself at distance 0
        pd_, pi_ = eng.search(probe, 2)
        # the same search through the fp16 pre-screen + fp32 rescoring: all 10^7 result entries must be the same bits
        eng.set_prescreen(True)
        eng.set_base(base)
        pdist, pidx = eng.search(queries, k)
        pst = eng.stats()
        eng.set_prescreen(False)
    assert pst["prescreen"] == 1 and pst["overflowed_queries"] == 0
    assert torch.equal(pidx, idx) and torch.equal(pdist.view(torch.int32), dist.view(torch.int32))
    del pdist, pidx
    assert st["overflowed_queries"] == 0 and st["tile_launches"] == 16      # 512 dense, x4 up to 512k rows seen, then ten 1M-row chunks
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())                                    # ascending (merge.py:104)
    assert int(idx.min()) >= 0 and int(idx.max()) < B
    srt = torch.sort(idx, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                                       # no duplicate neighbours
    assert (pi_[:, 0].cpu().numpy() == np.arange(9_876_543, 9_876_543 + 16)).all() and (pd_[:, 0].cpu().numpy() == 0).all()
    rows = np.linspace(0, Q - 1, 32).astype(np.int64)
    hq = queries[torch.from_numpy(rows).cuda()].cpu().numpy()
    di, ii = dist[torch.from_numpy(rows).cuda()].cpu().numpy(), idx[torch.from_numpy(rows).cuda()].cpu().numpy()
    del dist, idx, queries
    hb = base.cpu().numpy()                                                               # 41 GB D2H
    del base
    rd, ri = oracle.knn_canonical(hb, hq, k)
    assert np.array_equal(ii, ri)
    assert np.array_equal(di.view(np.uint32), rd.view(np.uint32))


@pytest.mark.timeout(1500)
def test_c4_per_gpu_share_properties_and_sampled_parity(oracle):
    """BASELINE.json config 4 as ONE of its eight GPUs sees it under base sharding: fp16 inputs, ALL 1M queries against its 12.5M-row
    shard of the 100M, d=1536, k=100 (operator cu_knn.py:243-245, global ids by + batch_offset, cu_knn.py:273).  At this size a filtered
    chunk is 3 907 query tiles x 4 096 base tiles = more workgroups than one HIP launch takes, so every chunk goes out as several launches
    (nwknn_api.cpp run_chunk), and the candidate lists are 8.4 GB.  Exact fp16 mode: the result is the canonical fp32 arithmetic applied
    to the fp16 values, so sampled rows must match the CPU oracle bit for bit against the ENTIRE shard; size-independent properties for
    all rows; a 20k-row slice of the queries searched on its own must give the same rows (no dependence on the query batch)."""
    import torch
    from neighborhoodwatch_amd import KnnEngine
    Q, B, d, k = 1_000_000, 12_500_000, 1536, 100
    offset = 37_500_000                                          # shard 3 of 8: global ids start here (+ batch_offset, cu_knn.py:273)
    with KnnEngine(0) as eng:
        base = eng.generate(oracle.SEED_BASE, offset, B, d, "f16")
        queries = eng.generate(oracle.SEED_QUERY, 0, Q, d, "f16")
        eng.set_base(base)                                       # fp16 input selects the fp16 compute path
        dist, idx = eng.search(queries, k, index_offset=offset)
        st = eng.stats()
        s0 = 654_321
        dist2, idx2 = eng.search(queries[s0:s0 + 20_000], k, index_offset=offset)
        probe = base[7_654_321:7_654_321 + 16].clone()
        pd_, pi_ = eng.search(probe, 2, index_offset=offset)
    assert st["prescreen"] == 1 and st["overflowed_queries"] == 0          # exact fp16 mode ran, no list overflowed
    assert st["prescreen_violations"] == 0 and st["prescreen_fallbacks"] == 0 and st["prescreen_checked"] > Q * k and st["prescreen_worst"] < 1.0
    # 18 chunks (512 dense, x4 up to 512k rows seen, then twelve 1M-row chunks) = 53 HIP launches at this query count: a chunk of 3 907 x 4 096
    # tiles exceeds one launch, so 1 + 5 + 2 (393k rows) + 11 x 4 + 2 (the last 441 600 rows)
    assert st["tile_launches"] == 53
    assert torch.equal(idx[s0:s0 + 20_000], idx2) and torch.equal(dist[s0:s0 + 20_000], dist2)
    bad_order = 0
    for r0 in range(0, Q, 100_000):                                          # in slabs: no second [Q,k] tensor at once
        dd, ii = dist[r0:r0 + 100_000], idx[r0:r0 + 100_000]
        bad_order += int((dd[:, 1:] < dd[:, :-1]).sum())                     # ascending (merge.py:104)
        srt = torch.sort(ii, dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())                       # no duplicate neighbours
    assert bad_order == 0
    assert int(idx.min()) >= offset and int(idx.max()) < offset + B          # global ids of this shard
    assert (pi_[:, 0].cpu().numpy() == offset + np.arange(7_654_321, 7_654_321 + 16)).all() and (pd_[:, 0].cpu().numpy() == 0).all()
    rows = np.linspace(0, Q - 1, 12).astype(np.int64)
    sel = torch.from_numpy(rows).cuda()
    hq = queries[sel].float().cpu().numpy()
    di, ii = dist[sel].cpu().numpy(), idx[sel].cpu().numpy()
    del dist, idx, dist2, idx2, queries
    hb = np.empty((B, d), dtype=np.float32)                                 # the fp16 VALUES as float32, converted on the GPU in slabs
    step = 1_250_000
    for s in range(0, B, step):
        hb[s:s + step] = base[s:s + step].float().cpu().numpy()
    del base
    rd, ri = oracle.knn_canonical(hb, hq, k)
    assert np.array_equal(ii, ri + offset)
    assert np.array_equal(di.view(np.uint32), rd.view(np.uint32))


@pytest.mark.timeout(1500)
def test_c5_maxsim_full_size_properties_and_sampled_parity(oracle):
    """BASELINE.json config 5 at full size on one GPU: 2 048 queries x 32 tokens against 1M documents x 64 tokens, d=128, k=1000
    (MaxSim is an extension: the reference's `ck` runs flat token kNN, colbert_knn.py:84-126, so the checker is the MaxSim oracle and
    there is no reference parity to claim).  Size-independent properties for all rows, pre-screened == direct bit for bit over the
    whole [2048, 1000] result, and 8 sampled queries recomputed by oracle.maxsim_canonical against ALL 1M documents."""
    import torch
    from neighborhoodwatch_amd import KnnEngine
    Q, B, d, k = 2_048, 1_000_000, 128, 1000
    with KnnEngine(0) as eng:
        docs = eng.generate(oracle.SEED_BASE, 0, B * 64, d).view(B, 64, d)
        queries = eng.generate(oracle.SEED_QUERY, 0, Q * 32, d).view(Q, 32, d)
        eng.maxsim_set_docs(docs)
        score, doc = eng.maxsim_search(queries, k)
        st = eng.stats()
        # a query made of a document's own first 32 tokens: every token finds itself, so the score is at least the sum of those
        # tokens' squared norms (~1 365 on this data, where an unrelated document scores ~300)
        probe_docs = torch.arange(654_321, 654_321 + 8, device="cuda")
        probe = docs[probe_docs, :32, :].clone()
        ps, pdoc = eng.maxsim_search(probe, 5)
        eng.set_prescreen(True)
        eng.maxsim_set_docs(docs)
        pscore, pdoc_all = eng.maxsim_search(queries, k)
        pst = eng.stats()
        eng.set_prescreen(False)
    assert st["prescreen"] == 0 and pst["prescreen"] == 1 and st["overflowed_queries"] == 0
    assert torch.equal(pdoc_all, doc) and torch.equal(pscore.view(torch.int32), score.view(torch.int32))     # pre-screened == direct, all 2 048 000 entries
    del pscore, pdoc_all
    assert bool((score[:, 1:] <= score[:, :-1]).all())                          # descending by score
    assert int(doc.min()) >= 0 and int(doc.max()) < B
    srt = torch.sort(doc, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                              # no duplicate documents
    self_norms = (probe.double() ** 2).sum(dim=2).sum(dim=1)                    # sum over the 32 tokens of |token|^2
    assert bool((ps[:, 0].double() >= self_norms * (1 - 1e-5)).all())
    assert bool((pdoc[:, 0] == probe_docs).all())                               # on random tokens nothing comes near a document's own prefix
    rows = np.linspace(0, Q - 1, 8).astype(np.int64)
    sel = torch.from_numpy(rows).cuda()
    hq = queries[sel].cpu().numpy()
    hs, hd = score[sel].cpu().numpy(), doc[sel].cpu().numpy()
    del score, doc, queries
    hdocs = docs.cpu().numpy()                                                   # 32.8 GB D2H
    del docs
    rs, ri = oracle.maxsim_canonical(hdocs, hq, k)
    assert np.array_equal(hd, ri)
    assert np.array_equal(hs.view(np.uint32), rs.view(np.uint32))
    assert np.array_equal(hdocs[777_000, :2], oracle.gen_numpy(oracle.SEED_BASE, 777_000 * 64, 2, d))
