"""Threshold metrics, AUROC/AUPRC and the hyper-parameter search over (beta, gamma, tau_*).

Host-side mirror of the pieces of lib/metrics/utils.py that run_lemon.py:319-427 drives:
  prob_metrics :408-412 | optimize_f1 :273-284 | optimize_f1_efficient :286-296 |
  f1_with_pred_prev_constraint(2) :298-322 | f1_with_local_minima_finder :327-349 |
  binary_metrics :351-405 | eval_metrics :414-441 | maximize_metric :151-196 (+ :84-149 helpers)
Same algorithms and the same scipy/sklearn entry points as the reference (so thresholds agree), but
every evaluation of the score goes through ONE callable `score_fn(hparams) -> scores[n]` which the
product binds to the K5 device kernel (ops.lemon_score) over arrays that stay in HBM; F1 is computed
from counts instead of sklearn.f1_score inside the inner loops (same value, ~50x cheaper).
Pinned by tests/golden/metrics.npz (generated from the reference).
"""
from itertools import product

import numpy as np
from scipy.optimize import bisect, fminbound, minimize
from scipy.signal import argrelextrema
from scipy.stats import gaussian_kde

HP_NAMES = ("beta", "gamma", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m")


def f1_binary(y, pred):
    """sklearn.metrics.f1_score(y, pred) for binary labels (0 when there is nothing to count)."""
    y = np.asarray(y).astype(bool)
    pred = np.asarray(pred).astype(bool)
    tp = int(np.count_nonzero(y & pred))
    denom = 2 * tp + int(np.count_nonzero(~y & pred)) + int(np.count_nonzero(y & ~pred))
    return 2.0 * tp / denom if denom else 0.0


def prob_metrics(targets, preds, sample_weight=None):
    from sklearn.metrics import average_precision_score, roc_auc_score
    return {"AUROC": roc_auc_score(targets, preds, sample_weight=sample_weight),
            "AUPRC": average_precision_score(targets, preds, average="macro", sample_weight=sample_weight)}


def optimize_f1(y, score, return_thres=False):
    """100-point threshold sweep between min and max score; ties go to the LARGER threshold."""
    score = np.asarray(score)
    best_thres, best_f1 = 0, 0
    for cand in np.linspace(score.min(), score.max(), 100):
        f1 = f1_binary(y, score >= cand)
        if f1 >= best_f1:
            best_f1, best_thres = f1, cand
    return (best_f1, best_thres) if return_thres else best_f1


def optimize_f1_efficient(y, score, return_thres=False):
    """Brent search (scipy fminbound, xtol 1e-8) on -F1(threshold); what the hparam search maximises."""
    score = np.asarray(score)
    neg = lambda t: -f1_binary(y, score >= t)
    best_thres = fminbound(neg, score.min(), score.max(), xtol=1e-8, disp=0)
    best_f1 = -neg(best_thres)
    return (best_f1, best_thres) if return_thres else best_f1


def f1_with_pred_prev_constraint2(y, score, pred_prev, return_thres=False):
    score = np.asarray(score)
    gap2 = lambda t: ((score >= t).sum() / len(score) - pred_prev) ** 2
    thres = fminbound(gap2, score.min(), score.max())
    f1 = f1_binary(y, score >= thres)
    return (f1, thres) if return_thres else f1


def f1_with_pred_prev_constraint(y, score, pred_prev, return_thres=False):
    """threshold at which the predicted prevalence equals `pred_prev` (bisection, with the squared-gap
    fminbound fallback when the bracket has no sign change or the result is NaN)."""
    score = np.asarray(score)
    gap = lambda t: (score >= t).sum() / len(score) - pred_prev
    try:
        thres = bisect(gap, score.min(), score.max())
        f1 = f1_binary(y, score >= thres)
    except ValueError:
        return f1_with_pred_prev_constraint2(y, score, pred_prev, return_thres)
    if np.isnan(thres) or np.isnan(f1):
        return f1_with_pred_prev_constraint2(y, score, pred_prev, return_thres)
    return (f1, thres) if return_thres else f1


def f1_with_local_minima_finder(y, score, return_thres=False):
    """threshold at the median local minimum of a Gaussian KDE of the scores (1000-point grid)."""
    score = np.asarray(score)
    xs = np.linspace(score.min(), score.max(), 1000)
    dens = gaussian_kde(score).evaluate(xs)
    minima = xs[argrelextrema(dens, np.less)]
    if len(minima) > 1:
        thres = np.median(minima)
    elif len(minima) == 1:
        thres = minima[0]
    else:
        maxima = xs[argrelextrema(dens, np.greater)]
        thres = np.median(maxima) if len(maxima) >= 2 else np.mean(score)
    f1 = f1_binary(y, score >= thres)
    return (f1, thres) if return_thres else f1


def binary_metrics(targets, preds, label_set=(0, 1), suffix="", return_arrays=False):
    from sklearn.metrics import (accuracy_score, balanced_accuracy_score, confusion_matrix, f1_score,
                                 recall_score)
    targets, preds = np.asarray(targets), np.asarray(preds)
    if len(targets) == 0:
        return {}
    res = {"accuracy": accuracy_score(targets, preds), "F1": f1_score(targets, preds), "n_samples": len(targets)}
    cm = confusion_matrix(targets, preds, labels=list(label_set))
    if len(label_set) == 2:
        tn, fp, fn, tp = (int(cm[0][0]), int(cm[0][1]), int(cm[1][0]), int(cm[1][1]))
        res.update(TN=tn, FN=fn, TP=tp, FP=fp, error=fn + fp)
        res["TPR"], res["FNR"] = ((tp / (tp + fn), fn / (tp + fn)) if tp + fn else (0, 1))
        res["FPR"], res["TNR"] = ((fp / (fp + tn), tn / (fp + tn)) if fp + tn else (1, 0))
        res["PPV"] = tp / (tp + fp) if tp + fp > 0 else 0
        res["NPV"] = tn / (tn + fn) if tn + fn > 0 else 0
        res["pred_prevalence"] = (tp + fp) / res["n_samples"]
        res["prevalence"] = (tp + fn) / res["n_samples"]
    else:
        res["TPR"] = recall_score(targets, preds, labels=list(label_set), average="macro", zero_division=0.0)
    if len(np.unique(targets)) > 1:
        res["balanced_acc"] = balanced_accuracy_score(targets, preds)
    if return_arrays:
        res["targets"], res["preds"] = targets, preds
    return {f"{k}{suffix}": v for k, v in res.items()}


def eval_metrics(y, score, prevalence, fix_thress=None, use_efficient=False):
    """AUROC/AUPRC + three thresholds (F1-optimal, prevalence-matched, KDE heuristic) and the binary
    metrics at each; thresholds can be frozen (`fix_thress`, chosen on val and re-used on test)."""
    fix_thress = fix_thress or {}
    y, score = np.asarray(y), np.asarray(score)
    if "F1_optimal_thres" in fix_thress:
        t_opt = fix_thress["F1_optimal_thres"]
    else:
        t_opt = (optimize_f1_efficient if use_efficient else optimize_f1)(y, score, True)[1]
    t_prev = fix_thress["F1_prev_thres"] if "F1_prev_thres" in fix_thress else \
        f1_with_pred_prev_constraint(y, score, prevalence, True)[1]
    t_heur = fix_thress["F1_heuristic_thres"] if "F1_heuristic_thres" in fix_thress else \
        f1_with_local_minima_finder(y, score, True)[1]
    return {**prob_metrics(y, score),
            "F1_optimal_thres": t_opt, "F1_prev_thres": t_prev, "F1_heuristic_thres": t_heur,
            **binary_metrics(y, score >= t_opt, suffix="_optimal"),
            **binary_metrics(y, score >= t_prev, suffix="_prev"),
            **binary_metrics(y, score >= t_heur, suffix="_heuristic")}


# ------------------------------------------------------------------------------ hyper-parameter search
def combinations_base(grid):
    return [dict(zip(grid.keys(), vals)) for vals in product(*grid.values())]


def unpack_vector(x, force_zero=(), force_one=()):
    cand = {name: x[i] for i, name in enumerate(HP_NAMES)}
    for name in cand:
        if name in force_zero:
            cand[name] = 0.0
    for name in cand:
        if name in force_one:
            cand[name] = 1.0
    return cand


def _torch_lbfgs(rec_t, y, x0, force_zero, force_one, max_iter=20):
    """SoftMargin proxy minimised by LBFGS (lib/metrics/utils.py:121-141,148-149): F1 is not
    differentiable, so the reference polishes a start point on a margin loss of the score.  Runs on
    whatever device rec_t lives on (the ~1100 closure evaluations of the four starts take 23 s on the CPU
    for a 5000 x 50 val split, ~2 s on the GPU; the candidate it returns is then scored exactly)."""
    import torch
    dev = rec_t["D_n"].device
    x = torch.tensor(x0, dtype=torch.float64, requires_grad=True, device=dev)
    opt = torch.optim.LBFGS([x], lr=0.1, max_iter=max_iter, line_search_fn="strong_wolfe")
    target = torch.as_tensor(np.asarray(y), dtype=torch.float64).to(dev) * 2 - 1

    def scores(x):
        hp = unpack_vector(x, force_zero, force_one)
        sn = torch.exp(-hp["tau_1_n"] * rec_t["D_n"]) * torch.exp(-hp["tau_2_n"] * rec_t["dists_tr_n"])
        sm = torch.exp(-hp["tau_1_m"] * rec_t["D_m"]) * torch.exp(-hp["tau_2_m"] * rec_t["dists_tr_m"])
        dn = (sn * rec_t["dists_n"]).sum(1) / rec_t["D_n"].shape[1]
        dm = (sm * rec_t["dists_m"]).sum(1) / rec_t["D_m"].shape[1]
        return rec_t["d_1"] + hp["beta"] * dn + hp["gamma"] * dm

    def closure():
        opt.zero_grad()
        loss = torch.nn.SoftMarginLoss()(scores(x), target)
        loss.backward()
        return loss

    for _ in range(max_iter):
        opt.step(closure)
    return x.detach().cpu().numpy()


def maximize_metric(score_fn, y, grid, x0s, obj_func=optimize_f1_efficient, obj_func_args=None,
                    force_zero=(), force_one=(), scipy_methods=("Powell", "Nelder-Mead"), rec_for_lbfgs=None,
                    batch_grid=None, lbfgs_device="cpu"):
    """lib/metrics/utils.py:151-196.  score_fn(hparams dict) -> scores (numpy [n]); y = is_mislabel.
    Order of candidates (and therefore tie-breaking on equal objective) follows the reference:
    scipy local searches from every start, LBFGS-polished starts, then the full grid; strict '>'."""
    obj_func_args = obj_func_args or {}
    y = np.asarray(y)

    def objective(x):
        hp = unpack_vector(x, force_zero, force_one)
        score = score_fn(hp)
        if not np.all(np.isfinite(score)):       # overflowing hyper-parameters: worst objective, not a crash
            return 0.0
        return -obj_func(y, score, **obj_func_args)

    best_x, best_val = None, -1
    for x0 in x0s:
        for method in scipy_methods:
            res = minimize(objective, x0, method=method, options={})
            if -res.fun > best_val:
                best_val, best_x = -res.fun, res.x
    if rec_for_lbfgs is not None:
        import torch
        rec_t = {k: torch.as_tensor(np.asarray(v), dtype=torch.float64 if k == "d_1" else torch.float32).to(lbfgs_device)
                 for k, v in rec_for_lbfgs.items()}
        for x0 in x0s:
            cand = _torch_lbfgs(rec_t, y, x0, force_zero, force_one)
            if not np.all(np.isfinite(cand)):
                # LBFGS on the SoftMargin proxy can diverge to NaN (seen on tiny synthetic sets); the reference
                # would die in fminbound with "bounds must be finite" (utils.py:170-174) -- skip the candidate
                continue
            val = -objective(cand)
            if val > best_val:
                best_val, best_x = val, cand
    gs = []
    for c in combinations_base(grid):
        g = []
        for name in HP_NAMES:
            if name in c:
                g.append(c[name])
            elif name in ("tau_1_n", "tau_1_m"):
                g.append(c["tau_1"])
            elif name in ("tau_2_n", "tau_2_m"):
                g.append(c["tau_2"])
            else:
                raise NotImplementedError(name)
            if name in force_zero:
                g[-1] = 0.0
        gs.append(g)
    # batch_grid: all grid points in one device launch (ops.grid_f1: same scores, same bounded-Brent search,
    # bit-identical values); the winner is still picked in grid order with the reference's strict '>'
    vals = batch_grid([unpack_vector(g, force_zero, force_one) for g in gs]) if (batch_grid is not None and gs) else None
    for j, g in enumerate(gs):
        val = float(vals[j]) if vals is not None else -objective(g)
        if val > best_val:
            best_val, best_x = val, g
    best_x = list(best_x)
    for i, name in enumerate(HP_NAMES):
        if name in force_zero:
            best_x[i] = 0.0
        if name in force_one:
            best_x[i] = 1.0
    final = score_fn(unpack_vector(best_x, force_zero, force_one))
    return best_x, best_val, obj_func(y, final, return_thres=True, **obj_func_args)[1]
