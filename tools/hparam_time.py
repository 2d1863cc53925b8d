"""Time the hyper-parameter search (run_lemon.py:319-384 protocol) on a CIFAR-scale val split (diagnostic)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd import ops, metrics as M
N, k = 5000, 50
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.rand(*s, generator=g, device="cuda")
y = (rnd(N) < 0.4).cpu().numpy()
rec = {"d_1": (rnd(N) + torch.as_tensor(y, device="cuda") * 0.1).double(), "D_n": -rnd(N, k), "dists_tr_n": rnd(N, k), "dists_n": rnd(N, k),
       "D_m": -rnd(N, k), "dists_tr_m": rnd(N, k), "dists_m": rnd(N, k)}
score_fn = lambda hp: ops.lemon_score(rec, hp).cpu().numpy()
grid = {"beta": np.arange(0, 100.01, 5), "gamma": np.arange(0, 100.01, 5), "tau_1": [0, 1, 5, 10], "tau_2": [0, 1, 5, 10]}
t0 = time.perf_counter()
best, f1, thr = M.maximize_metric(score_fn, y, grid, [[0] * 6, [0.5] * 6, [1] * 6, [10] * 6], M.optimize_f1_efficient, {},
                                  scipy_methods=())      # grid part only
print("grid only:", round(time.perf_counter() - t0, 2), "s", best, f1, thr)
t0 = time.perf_counter()
best, f1, thr = M.maximize_metric(score_fn, y, grid, [[0] * 6, [0.5] * 6, [1] * 6, [10] * 6], M.optimize_f1_efficient, {}, scipy_methods=(),
                                  batch_grid=lambda hps: ops.grid_f1(rec, y, [[hp[n] for n in M.HP_NAMES] for hp in hps])[0])
print("grid only, batched on the GPU:", round(time.perf_counter() - t0, 3), "s", best, f1, thr)
t0 = time.perf_counter()
best, f1, thr = M.maximize_metric(score_fn, y, {"beta": [0], "gamma": [0], "tau_1": [0], "tau_2": [0]},
                                  [[0] * 6, [0.5] * 6, [1] * 6, [10] * 6], M.optimize_f1_efficient, {})
print("scipy local searches only:", round(time.perf_counter() - t0, 2), "s", best, f1, thr)
