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
