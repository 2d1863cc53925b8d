"""CPU: the reference-style baseline legs of bench.py (oracle/reference_loop.py: flat index by torch.mm + top-k in 128-query
batches, the per-sample Python loop of run_lemon.py:238-307, its vectorised twin) agree with each other and with the C oracle."""
import numpy as np
import pytest
import torch

from oracle import reference_loop as rl


def _case(metric, seed=0, n_tr=700, n_all=900, d=32):
    g = torch.Generator().manual_seed(seed)
    img = torch.nn.functional.normalize(torch.randn(n_all, d, generator=g), dim=1)
    txt = torch.nn.functional.normalize(torch.randn(n_all, d, generator=g), dim=1)
    rng = np.random.default_rng(seed)
    in_compr = np.sort(rng.choice(n_all, n_tr, replace=False))       # run_lemon.py:123: the DB is a subset of the train split
    img_tr, txt_tr = img[in_compr], txt[in_compr]
    dists_tr = 1 - (txt_tr * img_tr).sum(1) if metric == "cosine" else ((txt_tr - img_tr) ** 2).sum(1)
    ii, it = rl.FlatIndexTorch(d, metric), rl.FlatIndexTorch(d, metric)
    ii.add(img_tr.numpy()); it.add(txt_tr.numpy())
    return img, txt, in_compr, img_tr, txt_tr, dists_tr, ii, it


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
@pytest.mark.parametrize("sname", ["train", "val"])
def test_loop_vectorised_and_oracle_agree(oracle, metric, sname):
    img, txt, in_compr, img_tr, txt_tr, dists_tr, ii, it = _case(metric)
    k, bs = 5, 128
    nq = 300
    q_img, q_txt = img[:nq], txt[:nq]
    logs = rl.per_sample_loop(sname, q_img, q_txt, img_tr, txt_tr, dists_tr, ii, it, k, bs, in_compr, metric)
    in_db = np.isin(np.arange(nq), in_compr)
    vec = rl.vectorised(sname, q_img.numpy(), q_txt.numpy(), img_tr.numpy(), txt_tr.numpy(), dists_tr.numpy(), ii, it, k, bs, in_db, metric)
    assert len(logs) == nq and [l["idx"] for l in logs] == list(range(nq))
    for key in ("I_n", "I_m"):
        assert np.array_equal(rl.stack(logs, key), vec[key])
    for key in ("d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
        assert np.allclose(rl.stack(logs, key), vec[key], rtol=0, atol=2e-6), key
    # the C oracle (float64 ranking, (distance, index) tie rule; self-exclusion by identity): NB in the loop a train-split
    # sample in the DB drops result[0] -- its own row, since the queries ARE the DB rows at those positions
    if sname == "train":
        # the oracle's drop_self semantics need query row == DB row for in-DB samples: true for the rows in in_compr
        sel = np.flatnonzero(in_db)
        db_pos = np.searchsorted(in_compr, sel)
        assert np.array_equal(img_tr.numpy()[db_pos], q_img.numpy()[sel])
    ref = oracle.neighbors(metric, img_tr.numpy(), txt_tr.numpy(), q_img.numpy(), q_txt.numpy(), k, drop_self=(sname == "train"),
                           in_db=in_db.astype(np.uint8) if sname == "train" else None)
    for key in ("I_n", "I_m"):
        assert np.array_equal(ref[key], vec[key]), key
    for key in ("d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
        assert np.allclose(ref[key], vec[key], rtol=0, atol=5e-6), key
