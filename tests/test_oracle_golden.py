"""CPU suite: pins the oracle (and the product's host logic) to golden vectors generated from the
importable reference modules (tools/make_golden.py), and to independent re-derivations."""
import ctypes
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ------------------------------------------------------------------ normalisation (lib/utils/utils.py:39-40)
def test_normalize_matches_reference_golden(oracle):
    g = load("normalize.npz")
    y = oracle.normalize_rows(g["x"])
    assert np.abs(y - g["y"]).max() <= 1.2e-7          # reference = torch fp32 reduction order
    assert (y[5] == 0).all()


# ------------------------------------------------------------------ score aggregation (lib/metrics/utils.py:21-82)
@pytest.mark.parametrize("k", [1, 5, 50])
def test_score_matches_reference_golden(oracle, k):
    g = load("scores.npz")
    hps = json.load(open(os.path.join(G, "scores_hparams.json")))
    rec = {nm: g[f"k{k}_{nm}"] for nm in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")}
    for hn, hp in hps.items():
        s, dn, dm = oracle.score(rec, hp, return_dn=True)
        # d_1 is float64 in the frame; the oracle takes it as float32 => 6e-8 absolute
        for got, key in ((s, "score"), (dn, "dn"), (dm, "dm"), (s, "score_loop")):
            ref = g[f"k{k}_{hn}_{key}"]
            assert np.allclose(got, ref, rtol=2e-6, atol=2e-7), (hn, key, np.abs(got - ref).max())


def test_auroc_matches_reference_golden(oracle):
    g = load("metrics.npz")
    rec = {nm: g[f"in_{nm}"] for nm in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")}
    hp = json.load(open(os.path.join(G, "scores_hparams.json")))["fixed"]
    s = oracle.score(rec, hp)
    assert np.allclose(s, g["score_fixed"], rtol=2e-6, atol=2e-7)
    assert abs(oracle.auroc(g["in_y"], g["score_fixed"]) - float(g["AUROC"])) < 1e-12
    assert abs(oracle.auroc(g["in_y"], s) - float(g["AUROC"])) < 5e-4   # "identical to 3 decimals" bar


# ------------------------------------------------------------------ kNN: independent second opinions
def _fmaf_ref(a, b):
    """dot(a,b) by libm's correctly-rounded fmaf, one call per step (pure Python loop: tiny sizes only)."""
    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    acc = 0.0
    for x, y in zip(a.tolist(), b.tolist()):
        acc = libm.fmaf(x, y, acc)
    return np.float32(acc)


def test_knn_chain_numerics_and_tie_rule(oracle):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((37, 19)).astype(np.float32)
    Q = rng.standard_normal((6, 19)).astype(np.float32)
    X[20] = X[3]; X[30] = X[3]                       # exact duplicates => ties
    D, I = oracle.knn("ip", X, Q, 9)
    S = np.array([[_fmaf_ref(q, x) for x in X] for q in Q], dtype=np.float32)
    for qi in range(6):
        order = sorted(range(37), key=lambda j: (-S[qi, j], j))[:9]
        assert I[qi].tolist() == order
        assert np.array_equal(D[qi], S[qi, order])
    D2, I2 = oracle.knn("l2", X, Q, 9)
    xn = np.array([_fmaf_ref(x, x) for x in X], dtype=np.float32)
    for qi in range(6):
        qn = _fmaf_ref(Q[qi], Q[qi])
        dist = np.maximum(np.float32(0), np.array(
            [np.float32(np.float64(np.float32(qn + xn[j])) - 2.0 * np.float64(S[qi, j])) for j in range(37)],
            dtype=np.float32))
        order = sorted(range(37), key=lambda j: (dist[j], j))[:9]
        assert I2[qi].tolist() == order
        assert np.allclose(D2[qi], dist[order], rtol=0, atol=1e-6)


def test_knn_against_reference_cosdistance_topk(oracle):
    # lib/metrics/utils.py:198-214 (in-tree torch brute force), untied random data: same neighbour sets
    g = load("cosdistance_topk.npz")
    f = oracle.normalize_rows(g["feat"])
    D, I = oracle.knn("ip", f, f, 6)
    assert np.array_equal(I, g["idx"])
    assert np.abs((1.0 - D) - g["vals"]).max() < 5e-7


def check_second_opinion(knn, normalize):
    """knn(metric, X, Q, k) -> (D, I) against tests/golden/knn_second_opinion.npz (tools/make_golden_knn.py): the
    reference's cosDistance + topk (IP) and its sklearn euclidean metric + argsort (L2) at N = 2 048, d = 64, k = 51.
    Values must agree everywhere; index rows must agree wherever the reference's own consecutive values are further
    apart than float32 noise (two float32 implementations may order a 1-ulp near-tie differently)."""
    from tests.synth import second_opinion_inputs
    g = load("knn_second_opinion.npz")
    feat, X, Q = second_opinion_inputs()
    f = normalize(feat)
    D, I = knn("ip", f, np.ascontiguousarray(f[::4]), 51)
    assert np.abs((1.0 - D) - g["ip_vals"]).max() < 1e-6             # (sgemm vs fma chain over d = 64; north_star: 1e-4)
    clear = np.diff(g["ip_vals"], axis=1).min(1) > 1e-6
    assert clear.mean() > 0.9 and np.array_equal(I[clear], g["ip_idx"][clear].astype(np.int64))
    assert (I != g["ip_idx"]).sum() <= 2 * (~clear).sum()           # a near-tie swaps two neighbouring entries
    D, I = knn("l2", X, Q, 51)
    assert np.array_equal(I, g["l2_idx"].astype(np.int64))           # smallest gap 1.5e-5: no near-ties
    assert np.abs(np.sqrt(D) - g["l2_dist"]).max() < 2e-5 * g["l2_dist"].max()


def test_knn_against_reference_second_opinion_n2048_k51(oracle):
    check_second_opinion(oracle.knn, oracle.normalize_rows)


def test_knn_against_torch_mm_topk(oracle):
    rng = np.random.default_rng(1)
    X = oracle.normalize_rows(rng.standard_normal((3000, 96)).astype(np.float32))
    Q = oracle.normalize_rows(rng.standard_normal((200, 96)).astype(np.float32))
    D, I = oracle.knn("ip", X, Q, 10)
    S = torch.from_numpy(Q).double() @ torch.from_numpy(X).double().T
    v, i = S.topk(10, dim=1)
    assert (i.numpy() == I).mean() > 0.999             # fp32-vs-fp64 boundary flips only
    assert np.abs(v.numpy() - D).max() < 1e-6
    D2, I2 = oracle.knn("l2", X, Q, 10)
    assert (I2 == I).mean() > 0.995 and np.abs(D2 - (2 - 2 * D)).max() < 2e-6


def test_knn_padding_and_edges(oracle):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((3, 8)).astype(np.float32)
    Q = rng.standard_normal((2, 8)).astype(np.float32)
    D, I = oracle.knn("ip", X, Q, 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == -np.finfo(np.float32).max).all()
    D, I = oracle.knn("l2", X, Q, 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == np.finfo(np.float32).max).all()
    D, I = oracle.knn("ip", X[:0], Q, 2)
    assert (I == -1).all()


def test_neighbors_restates_the_reference_loop(oracle):
    """oracle.neighbors vs a literal numpy transcription of the per-sample loop semantics
    (Appendix A of SURVEY.md) built on oracle.knn only."""
    from tests.synth import planted
    s = planted(seed=3, n_tr=400, n_q=50, d=32, C=8)
    img_tr, txt_tr, _, noisy_tr = s["train"]
    k = 4
    for metric in ("cosine", "euclidean"):
        for drop_self in (False, True):
            if drop_self:
                q_img, q_txt, noisy_q = img_tr[:60], txt_tr[:60], noisy_tr[:60]
                in_db = (np.arange(60) % 5 != 0).astype(np.uint8)
            else:
                q_img, q_txt, _, noisy_q = s["query"]
                in_db = None
            for discrete in (False, True):
                out = oracle.neighbors(metric, img_tr, txt_tr, q_img, q_txt, k, drop_self, in_db, discrete,
                                       noisy_tr, noisy_q)
                ks = k + int(drop_self)
                Dn, In = oracle.knn(metric, img_tr, q_img, ks)
                Dm, Im = oracle.knn(metric, txt_tr, q_txt, ks)
                dists_tr = oracle.paired_distance(metric, txt_tr, img_tr)
                for i in range(len(q_img)):
                    sl = slice(0, k)
                    if drop_self:
                        sl = slice(1, ks) if in_db[i] else slice(0, ks - 1)
                    I_n, D_n, I_m, D_m = In[i][sl], Dn[i][sl], Im[i][sl], Dm[i][sl]
                    assert np.array_equal(out["I_n"][i], I_n) and np.array_equal(out["I_m"][i], I_m)
                    if discrete:
                        exp_dn = 1.0 - (noisy_tr[I_n] == noisy_q[i]).astype(np.float32)
                        exp_Dn = D_n
                    elif metric == "cosine":
                        exp_dn = np.array([1 - oracle.dot_chain(q_txt[i], txt_tr[j]) for j in I_n], np.float32)
                        exp_Dn = -D_n
                    else:
                        exp_dn = oracle.paired_distance(metric, np.repeat(q_txt[i:i + 1], k, 0), txt_tr[I_n])
                        exp_Dn = D_n
                    assert np.array_equal(out["dists_n"][i], exp_dn)
                    assert np.array_equal(out["D_n"][i], exp_Dn)
                    assert np.array_equal(out["D_m"][i], -D_m if metric == "cosine" else D_m)
                    assert np.array_equal(out["dists_tr_n"][i], dists_tr[I_n])
                    assert np.array_equal(out["dists_tr_m"][i], dists_tr[I_m])
                    exp_dm = oracle.paired_distance(metric, np.repeat(q_img[i:i + 1], k, 0), img_tr[I_m])
                    assert np.array_equal(out["dists_m"][i], exp_dm)


def test_planted_noise_is_detectable(oracle):
    """End-to-end sanity on S-small: LEMoN score has AUROC >> 0.5 and beats d_1 alone."""
    from tests.synth import planted
    s = planted(seed=0, n_tr=2048, n_q=256, d=64, C=16)
    img_tr, txt_tr, _, _ = s["train"]
    q_img, q_txt, clean, noisy = s["query"]
    rec = oracle.neighbors("cosine", img_tr, txt_tr, q_img, q_txt, 5)
    y = clean != noisy
    a1 = oracle.auroc(y, rec["d_1"])
    a2 = oracle.auroc(y, oracle.score(rec, dict(beta=5, gamma=5, tau_1_n=0.1, tau_2_n=5, tau_1_m=0.1, tau_2_m=5)))
    assert a1 > 0.8 and a2 > 0.8
