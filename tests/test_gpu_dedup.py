"""GPU: query de-duplication (csrc/dedup.hip) is exact and actually folds duplicates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
@pytest.mark.parametrize("nq,C", [(5000, 100), (1024, 1), (3000, 1400), (4096, 4096)])
def test_search_with_duplicate_queries_is_bit_identical(hip, oracle, metric, nq, C):
    from lemon_amd.index import IndexFlatIP, IndexFlatL2
    from tests.synth import unit_rows
    rng = np.random.default_rng(nq + C)
    d, n, k = 96, 3000, 11
    X = unit_rows(rng, n, d)
    X[1::7] = X[0]                                             # duplicate DB rows too: ties go to the lower index
    protos = unit_rows(rng, C, d)
    protos[0] = X[5]
    assign = rng.integers(0, C, nq)
    assign[:C] = np.arange(C)                                  # every prototype occurs
    Q = np.ascontiguousarray(protos[assign])
    cls = IndexFlatIP if metric == "cosine" else IndexFlatL2
    a, b = cls(d), cls(d)
    b.set_query_dedup(False)
    xt, qt = torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda()
    a.add(xt); b.add(xt)
    Da, Ia = a.search(qt, k)
    Db, Ib = b.search(qt, k)
    assert torch.equal(Ia, Ib) and torch.equal(Da, Db)
    info_a, info_b = a.last_search_info(), b.last_search_info()
    n_distinct = len({r.tobytes() for r in Q})
    assert info_b["nq_distinct"] == nq
    assert info_a["nq_distinct"] == (n_distinct if 2 * n_distinct <= nq else nq)
    Do, Io = oracle.knn(metric, X, Q[:300], k)
    assert np.array_equal(Ia[:300].cpu().numpy(), Io) and np.array_equal(Da[:300].cpu().numpy(), Do)


def test_rows_differing_in_one_bit_are_not_merged(hip):
    from lemon_amd.index import IndexFlatIP
    rng = np.random.default_rng(0)
    d = 64
    base = rng.standard_normal(d).astype(np.float32)
    Q = np.tile(base, (2048, 1))
    Q[1000, 63] = np.nextafter(Q[1000, 63], np.float32(10))   # one ulp in the last column
    Q[7, 0] = -Q[7, 0]
    X = rng.standard_normal((500, d)).astype(np.float32)
    ix = IndexFlatIP(d)
    ix.add(torch.from_numpy(X).cuda())
    ix.search(torch.from_numpy(Q).cuda(), 3)
    assert ix.last_search_info()["nq_distinct"] == 3


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
@pytest.mark.parametrize("k", [65, 100, 200])
def test_deep_k_equals_oracle_and_extends_shallow_search(hip, oracle, metric, k):
    """k > 64 (faiss has no limit): key-bounded passes of the exact scan.  The deep list equals the oracle's bit for bit,
    its first 64 columns equal a plain k=64 search, ties (duplicate rows) keep ascending index across pass boundaries,
    and k > ntotal pads with -1 / +-FLT_MAX."""
    from lemon_amd.index import IndexFlatIP, IndexFlatL2
    from tests.synth import unit_rows
    rng = np.random.default_rng(k)
    d, n, nq = 48, 700, 37
    X = unit_rows(rng, n, d)
    X[100:240] = X[100]                         # 140 exact duplicates: a tie group straddling the 64 / 128 boundaries
    Q = unit_rows(rng, nq, d)
    Q[0] = X[100]
    ix = (IndexFlatIP if metric == "cosine" else IndexFlatL2)(d)
    ix.add(torch.from_numpy(X).cuda())
    D, I = ix.search(torch.from_numpy(Q).cuda(), k)
    Do, Io = oracle.knn(metric, X, Q, k)
    assert np.array_equal(I.cpu().numpy(), Io) and np.array_equal(D.cpu().numpy(), Do)
    D64, I64 = ix.search(torch.from_numpy(Q).cuda(), 64)
    assert torch.equal(I[:, :64], I64) and torch.equal(D[:, :64], D64)
    small = (IndexFlatIP if metric == "cosine" else IndexFlatL2)(d)
    small.add(torch.from_numpy(X[:90]).cuda())
    Ds, Is = small.search(torch.from_numpy(Q).cuda(), k)
    Dso, Iso = oracle.knn(metric, X[:90], Q, k)
    assert np.array_equal(Is.cpu().numpy(), Iso) and np.array_equal(Ds.cpu().numpy(), Dso)
    assert (Is[:, 90:] == -1).all()


def test_index_data_view_and_reconstruct(hip):
    from lemon_amd.index import IndexFlatIP
    x = torch.randn(50, 24, device="cuda")
    ix = IndexFlatIP(24)
    assert ix.data().shape == (0, 24)
    ix.add(x)
    assert torch.equal(ix.data(), x) and torch.equal(ix.reconstruct_n(10, 5), x[10:15])
