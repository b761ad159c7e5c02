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
    assert st["overflowed_queries"] == 0 and st["tile_launches"] == 21
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
