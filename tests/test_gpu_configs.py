"""GPU: BASELINE.json configs[2] and configs[4] at their FULL shapes, through size-independent properties
(the oracle finishes only a few hundred rows of these in seconds)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(n, d, seed, dev):
    from lemon_amd import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty((n, d), device=dev)
    for i in range(0, n, 1 << 18):
        out[i:i + (1 << 18)].normal_(generator=g)
    return ops.normalize_vectors(out)


def test_config2_mscoco_shape_caption_side_knn(hip, oracle):
    """configs[2]: 82 783 train + 5 000 val + 5 000 test = 92 783 queries against a random 50 000-row DB
    subset of train (run_lemon.py:122-124), 512-d, UNIQUE caption embeddings, train rows mostly NOT in the DB."""
    from lemon_amd import _lib
    from lemon_amd.neighbors import LemonDB
    dev = torch.device("cuda", 0)
    n_tr, n_va, n_te, n_db, d, k = 82783, 5000, 5000, 50000, 512, 5
    img, txt = _unit(n_tr + n_va + n_te, d, 1, dev), _unit(n_tr + n_va + n_te, d, 2, dev)
    sel = torch.from_numpy(np.random.RandomState(0).choice(n_tr, n_db, replace=False)).to(dev)
    in_db = torch.zeros(n_tr + n_va + n_te, dtype=torch.uint8, device=dev)
    in_db[sel] = 1
    db = LemonDB(img[sel], txt[sel], "cosine")
    rec = db.neighbors(img, txt, k, drop_self=True, in_db=in_db)       # val/test ride along with in_db = 0 (pipeline.score_splits)
    info = db.index_txt.last_search_info()
    assert info["nq_distinct"] == n_tr + n_va + n_te                    # captions are unique: nothing folds
    # (a) every algorithm gives the same bits on a strided subset
    sub = torch.arange(0, n_tr + n_va + n_te, 23, device=dev)
    for algo in (_lib.ALGO_F32_MFMA, _lib.ALGO_BF16_FILTER):
        db2 = LemonDB(img[sel], txt[sel], "cosine", algo=algo)
        r2 = db2.neighbors(img[sub], txt[sub], k, drop_self=True, in_db=in_db[sub])
        for key in ("I_n", "I_m", "D_n", "D_m", "dists_n", "dists_m", "dists_tr_n", "dists_tr_m", "d_1"):
            assert torch.equal(rec[key][sub], r2[key]), (algo, key)
    # (b) the oracle on 200 spread rows (train rows inside and outside the DB, val, test)
    rows = torch.cat([torch.arange(0, n_tr, n_tr // 150, device=dev)[:150], torch.arange(n_tr, n_tr + 50, device=dev)])
    ref = oracle.neighbors("cosine", img[sel].cpu().numpy(), txt[sel].cpu().numpy(), img[rows].cpu().numpy(),
                           txt[rows].cpu().numpy(), k, drop_self=True, in_db=in_db[rows].cpu().numpy())
    for key in ("I_n", "I_m", "D_n", "D_m", "dists_n", "dists_m", "dists_tr_n", "dists_tr_m", "d_1"):
        assert np.array_equal(rec[key][rows].cpu().numpy(), ref[key]), key
    # (c) self-exclusion property at full size: a train row that IS in the DB never lists itself, and its best
    #     image neighbour is not better than itself would have been
    pos = torch.full((n_tr + n_va + n_te,), -1, dtype=torch.int64, device=dev)
    pos[sel] = torch.arange(n_db, device=dev)
    m = in_db.bool()
    assert not (rec["I_n"][m] == pos[m][:, None]).any() and not (rec["I_m"][m] == pos[m][:, None]).any()
    assert (rec["D_n"] <= 0).all() and (rec["D_n"][:, :-1] <= rec["D_n"][:, 1:]).all()      # -IP, best first


def test_config4_cc3m_scale_self_join(hip, oracle):
    """configs[4] shape: 3 M x 768 embeddings, k = 30 (experiments.py:253), full-DB self-join on ONE GPU: the bf16
    filter scan over all 3 M queries (6 query chunks x 36 database chunks with carried state) must equal the exact
    fp32 scan bit for bit on 8 192 strided queries, and the oracle on 64 of those."""
    from lemon_amd import _lib
    from lemon_amd.index import IndexFlatIP
    dev = torch.device("cuda", 0)
    n, d, k = 3_000_000, 768, 31
    X = _unit(n, d, 7, dev)
    a = IndexFlatIP(d)
    a.set_algo(_lib.ALGO_BF16_FILTER)
    a.add(X)
    D, I = a.search(X, k)
    torch.cuda.synchronize()
    assert a.last_search_info()["algo"] == _lib.ALGO_BF16_FILTER
    assert (I[:, 0] == torch.arange(n, device=dev)).float().mean() > 0.999999      # self is rank 0 (unit rows)
    assert (D[:, :-1] >= D[:, 1:]).all()
    sub = torch.arange(0, n, n // 8192, device=dev)[:8192]
    b = IndexFlatIP(d)
    b.set_algo(_lib.ALGO_F32_MFMA)
    b.add(X)
    Db, Ib = b.search(X[sub], k)
    assert torch.equal(I[sub], Ib) and torch.equal(D[sub], Db)
    del a, b
    rows = sub[::128]
    Do, Io = oracle.knn("ip", X.cpu().numpy(), X[rows].cpu().numpy(), k)
    assert np.array_equal(I[rows].cpu().numpy(), Io) and np.array_equal(D[rows].cpu().numpy(), Do)


@pytest.mark.parametrize("nq,n,d,k", [(70000, 1500, 64, 10),      # 547 panels x 12 tiles: whole panels per workgroup + a shared pool
                                      (9000, 300, 32, 7),         # 71 panels x 3 tiles: fewer workgroups than slots
                                      (33000, 9000, 128, 20),     # 258 panels x 71 tiles
                                      (1000, 70000, 64, 10),      # 8 panels x 547 tiles: one group, many pieces per panel
                                      (20000, 33000, 96, 33)])    # 157 panels x 258 tiles, d not a multiple of 64
def test_exact_scan_plan_regimes_against_the_oracle_on_sampled_queries(hip, oracle, nq, n, d, k):
    """The planned decomposition of the exact scan (lemon_plan_segments: heads / tails / whole panels / pool, pieces merged
    by k_merge) at shapes that exercise every branch of the planner: the rows of 512 sampled queries are bit-identical
    to the oracle's, every row is sorted, and nothing changes when the same queries are searched in a different batch
    (another plan)."""
    import lemon_amd
    dev = torch.device("cuda", 0)
    x, q = _unit(n, d, 11, dev), _unit(nq, d, 12, dev)
    idx = lemon_amd.IndexFlatIP(d)
    idx.set_algo(1)                                   # the exact fp32 scan
    idx.add(x)
    D, I = idx.search(q, k)
    torch.cuda.synchronize()
    assert (D[:, :-1] >= D[:, 1:]).all() and (I >= 0).all() and (I < n).all()
    sel = torch.from_numpy(np.random.RandomState(3).choice(nq, 512, replace=False)).to(dev)
    Do, Io = oracle.knn("ip", x.cpu().numpy(), q[sel].cpu().numpy(), k)
    assert np.array_equal(I[sel].cpu().numpy(), Io) and np.array_equal(D[sel].cpu().numpy(), Do)
    D2, I2 = idx.search(q[sel], k)                    # 4 panels: a different plan, same rows
    assert torch.equal(I2, I[sel]) and torch.equal(D2, D[sel])
