"""CPU suite: lemon_amd.metrics (threshold metrics + hyper-parameter search) against the golden
vectors produced by the reference's lib/metrics/utils.py (tests/golden/metrics.npz)."""
import json
import os

import numpy as np
import pytest

from lemon_amd import metrics as M

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
g = np.load(os.path.join(G, "metrics.npz"), allow_pickle=False)
y, score = g["in_y"], g["score_fixed"]
REC = {nm: g[f"in_{nm}"] for nm in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")}


def test_prob_metrics_and_f1_thresholds():
    pm = M.prob_metrics(y, score)
    assert abs(pm["AUROC"] - float(g["AUROC"])) < 1e-12 and abs(pm["AUPRC"] - float(g["AUPRC"])) < 1e-12
    for fn, fk, tk in ((M.optimize_f1_efficient, "f1_eff", "thres_eff"), (M.optimize_f1, "f1_grid", "thres_grid"),
                       (M.f1_with_local_minima_finder, "f1_heur", "thres_heur")):
        f1, th = fn(y, score, True)
        assert abs(f1 - float(g[fk])) < 1e-12 and abs(th - float(g[tk])) < 1e-9, fn.__name__
    f1, th = M.f1_with_pred_prev_constraint(y, score, y.mean(), True)
    assert abs(f1 - float(g["f1_prev"])) < 1e-12 and abs(th - float(g["thres_prev"])) < 1e-9


def test_f1_binary_equals_sklearn():
    from sklearn.metrics import f1_score
    rs = np.random.RandomState(0)
    for _ in range(20):
        a, b = rs.rand(50) < 0.4, rs.rand(50) < 0.5
        assert abs(M.f1_binary(a, b) - f1_score(a, b)) < 1e-15
    assert M.f1_binary(np.zeros(5), np.zeros(5)) == 0.0


def test_eval_metrics_matches_reference():
    ev = M.eval_metrics(y, score, prevalence=y.mean())
    keys, vals = g["eval_keys"].tolist(), g["eval_vals"]
    assert sorted(k for k, v in ev.items() if np.isscalar(v)) == keys
    for k, v in zip(keys, vals):
        assert abs(float(ev[k]) - v) < 1e-9, k
    frozen = M.eval_metrics(y, score, prevalence=y.mean(), fix_thress=ev)
    assert frozen["F1_optimal"] == ev["F1_optimal"] and frozen["F1_prev_thres"] == ev["F1_prev_thres"]


def test_hparam_search_matches_reference(oracle):
    score_fn = lambda hp: oracle.score(REC, hp)
    grid = {"beta": [0, 5, 10], "gamma": [0, 5, 10], "tau_1": [0, 1], "tau_2": [0, 5]}
    # grid-only optimum: deterministic, what a batched device grid search must reproduce exactly
    bx, bv, bt = M.maximize_metric(score_fn, y, grid, [], scipy_methods=())
    assert abs(bv - float(g["grid_best_val"])) < 1e-12 and np.allclose(bx, g["grid_best_x"])
    # full protocol (local searches + LBFGS-polished starts + grid), same starts as the golden run
    bx, bv, bt = M.maximize_metric(score_fn, y, grid, [[0] * 6, [1] * 6], scipy_methods=("Nelder-Mead",),
                                   rec_for_lbfgs=REC)
    assert abs(bv - float(g["search_best_val"])) < 1e-9
    assert np.allclose(bx, g["search_best_x"], rtol=1e-6, atol=1e-8)
    assert abs(bt - float(g["search_best_thres"])) < 1e-6


def test_force_zero_one_and_unpack():
    hp = M.unpack_vector([1, 2, 3, 4, 5, 6], force_zero=["tau_1_n", "tau_1_m"], force_one=["beta"])
    assert hp == dict(beta=1.0, gamma=2, tau_1_n=0.0, tau_2_n=4, tau_1_m=0.0, tau_2_m=6)
    assert len(M.combinations_base({"a": [1, 2], "b": [3, 4, 5]})) == 6
