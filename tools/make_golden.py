"""Generates tests/golden/*.npz|json by IMPORTING the importable pieces of the reference
(/root/reference, present only in the build container) on fixed seeded inputs.

What is importable, and with which stubs, is recorded in SURVEY.md 8c:
  lib/metrics/utils.py        (fake netcal.metrics.ECE + fake lib.datasets.utils label arrays)
  lib/datasets/utils.py       (MagicMock torchvision*, lib.datasets.clustering)
  lib/datasets/noise_captioning.py, lib/utils/utils.py  (plain import)
run_lemon.py itself, faiss and the CLIP loaders are NOT importable here (missing modules /
no network); nothing is generated for them.

Outputs are DATA (inputs + expected outputs), never reference source text.  Also writes
lemon_amd/data/label_sets.json (dataset class-name metadata used by the product).
Run:  python tools/make_golden.py
"""
import hashlib
import importlib
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import pandas as pd

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def purge(prefix):
    for k in list(sys.modules):
        if k == prefix or k.startswith(prefix + "."):
            del sys.modules[k]


sys.path.insert(0, REF)

# ------------------------------------------------------------------ datasets/utils (labels, noise, splits)
for m in ("torchvision", "torchvision.transforms", "torchvision.datasets", "lib.datasets.clustering"):
    sys.modules[m] = MagicMock()
dsu = importlib.import_module("lib.datasets.utils")

labels = {
    "cifar10": dsu.cifar10_labels.tolist(), "cifar100": dsu.cifar100_labels.tolist(),
    "mini_imagenet": dsu.mini_imagenet_labels.tolist(), "stanford_cars": dsu.stanford_cars_labels.tolist(),
}
meta = {
    "labels": labels,
    "class_num_dict": {k: int(v) for k, v in dsu.class_num_dict.items()},
    "CLIP_MEAN": list(dsu.CLIP_MEAN), "CLIP_STD": list(dsu.CLIP_STD),
}
os.makedirs(os.path.join(ROOT, "lemon_amd", "data"), exist_ok=True)
with open(os.path.join(ROOT, "lemon_amd", "data", "label_sets.json"), "w") as f:
    json.dump(meta, f)
with open(os.path.join(OUT, "dataset_meta.json"), "w") as f:
    json.dump({**meta, "labels_sha256": {k: hashlib.sha256("\n".join(v).encode()).hexdigest()
                                         for k, v in labels.items()}}, f)

noise = {}
import contextlib, io
for ds, C in (("cifar10", 10), ("cifar100", 100)):
    y = (np.random.RandomState(123).randint(0, C, 3000)).astype(np.int64)
    noise[f"{ds}_y"] = y
    for seed in (0, 1, 2):
        for lvl in (0.2, 0.4):
            with contextlib.redirect_stdout(io.StringIO()):
                noise[f"{ds}_asymmetric_{seed}_{lvl}"] = dsu.add_noisy_labels(ds, "asymmetric", lvl, seed, list(y))
                noise[f"{ds}_symmetric_{seed}_{lvl}"] = dsu.add_noisy_labels(ds, "symmetric", lvl, seed, list(y))
try:
    dsu.add_noisy_labels("cifar100", "cat", 0.4, 0, list(noise["cifar100_y"]))
    noise["cat_raises"] = np.array(0)
except NotImplementedError:
    noise["cat_raises"] = np.array(1)
np.savez_compressed(os.path.join(OUT, "noise_labels.npz"), **noise)

from sklearn.model_selection import train_test_split  # what lib/datasets/utils.py:409-410 calls
splits = {}
for seed in (0, 1, 2):
    tr, va = train_test_split(np.arange(50000), test_size=0.2, random_state=seed)
    va, te = train_test_split(va, test_size=0.5, random_state=seed)
    if seed == 0:
        splits["train_0"], splits["val_0"], splits["test_0"] = tr.astype(np.int32), va.astype(np.int32), te.astype(np.int32)
    splits[f"sha_{seed}"] = np.array([sha(tr.astype(np.int64)), sha(va.astype(np.int64)), sha(te.astype(np.int64))])
    splits[f"head_{seed}"] = np.stack([tr[:16], va[:16], te[:16]]).astype(np.int32)
np.savez_compressed(os.path.join(OUT, "splits.npz"), **splits)

# ------------------------------------------------------------------ noise_captioning
nc = importlib.import_module("lib.datasets.noise_captioning")
cap = {}
for seed in (0, 7):
    d = nc.random_noise_dict(50, 0.4, seed)
    cap[f"random_{seed}_keys"] = np.array(list(d.keys()), dtype=np.int64)
    cap[f"random_{seed}_vals"] = np.array(list(d.values()), dtype=np.int64)
rs = np.random.RandomState(5)
cats = [sorted(set(rs.randint(0, 12, rs.randint(0, 4)).tolist())) for _ in range(60)]
cap["cats_flat"] = np.array([c for row in cats for c in row], dtype=np.int64)
cap["cats_len"] = np.array([len(r) for r in cats], dtype=np.int64)
for seed in (0, 3):
    d = nc.calc_noise_by_integer_matching(np.array(cats, dtype=object), 0.4, seed)
    cap[f"match_{seed}_keys"] = np.array(list(d.keys()), dtype=np.int64)
    cap[f"match_{seed}_vals"] = np.array(list(d.values()), dtype=np.int64)
frame = pd.DataFrame({"sentence": [f"caption {i % 37}" for i in range(60)]}, index=np.arange(100, 160))
d = nc.random_noise_dict(60, 0.3, 1)
noised = nc.noise_given_dict(frame, d)
cap["given_sentence_id"] = np.array([int(s.split()[1]) for s in noised["sentence"]], dtype=np.int64)
cap["given_is_mislabel"] = noised["is_mislabel"].values.astype(np.uint8)
np.savez_compressed(os.path.join(OUT, "noise_captioning.npz"), **cap)

# ------------------------------------------------------------------ lib/utils/utils.normalize_vectors
import torch
uu = importlib.import_module("lib.utils.utils")
rs = np.random.RandomState(0)
x = (rs.randn(64, 48) * rs.uniform(0.01, 20, (64, 1))).astype(np.float32)
x[5] = 0
np.savez_compressed(os.path.join(OUT, "normalize.npz"), x=x, y=uu.normalize_vectors(torch.from_numpy(x)).numpy())

# ------------------------------------------------------------------ lib/metrics/utils (aggregation, AUROC, F1, search)
purge("lib.datasets")
fake_ds = types.ModuleType("lib.datasets.utils")
fake_ds.cifar10_labels = np.array(labels["cifar10"])
fake_ds.cifar100_labels = np.array(labels["cifar100"])
pkg = types.ModuleType("lib.datasets")
pkg.utils = fake_ds
sys.modules["lib.datasets"] = pkg
sys.modules["lib.datasets.utils"] = fake_ds
netcal = types.ModuleType("netcal")
netcal_m = types.ModuleType("netcal.metrics")
netcal_m.ECE = type("ECE", (), {"measure": lambda self, a, b: 0.0})
sys.modules["netcal"] = netcal
sys.modules["netcal.metrics"] = netcal_m
mu = importlib.import_module("lib.metrics.utils")


def make_frame(rs, n, k, informative=True):
    y = (rs.rand(n) < 0.4).astype(np.int64)
    shift = 0.25 * y[:, None] if informative else 0.0
    rec = {
        "d_1": (rs.rand(n).astype(np.float32) * 0.3 + 0.6 + 0.1 * y).astype(np.float64),  # `d1.item()` => python float
        "D_n": -(rs.rand(n, k).astype(np.float32) * 0.5 + 0.4),
        "dists_tr_n": rs.rand(n, k).astype(np.float32) * 0.4 + 0.5,
        "dists_n": (rs.rand(n, k).astype(np.float32) * 0.5 + shift).astype(np.float32),
        "D_m": -(rs.rand(n, k).astype(np.float32) * 0.5 + 0.4),
        "dists_tr_m": rs.rand(n, k).astype(np.float32) * 0.4 + 0.5,
        "dists_m": (rs.rand(n, k).astype(np.float32) * 0.5 + shift).astype(np.float32),
    }
    df = pd.DataFrame({"d_1": rec["d_1"], "is_mislabel": y})
    for c in ("D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m"):
        df[c] = list(rec[c])
    return df, rec, y


HPS = {
    "zero": dict(beta=0, gamma=0, tau_1_n=0, tau_2_n=0, tau_1_m=0, tau_2_m=0),
    "fixed": dict(beta=5, gamma=5, tau_1_n=0.1, tau_2_n=5, tau_1_m=0.1, tau_2_m=5),   # train_clip_from_scratch.py:102-109
    "random": dict(beta=12.5, gamma=3.25, tau_1_n=1.0, tau_2_n=0.5, tau_1_m=10.0, tau_2_m=1.0),
    "npfloat": {k: np.float64(v) for k, v in dict(beta=20.0, gamma=45.0, tau_1_n=5.0, tau_2_n=1.0, tau_1_m=5.0, tau_2_m=1.0).items()},
}
sc = {}
rs = np.random.RandomState(42)
for k in (1, 5, 50):
    df, rec, y = make_frame(rs, 40, k)
    for nm, arr in rec.items():
        sc[f"k{k}_{nm}"] = arr
    for hn, hp in HPS.items():
        s, dn, dm = mu.calc_scores_given_hparams_vectorized(df, hp, True)
        s2 = mu.calc_scores_given_hparams(df, hp)
        sc[f"k{k}_{hn}_score"] = np.asarray(s, dtype=np.float64)
        sc[f"k{k}_{hn}_dn"] = np.asarray(dn, dtype=np.float64)
        sc[f"k{k}_{hn}_dm"] = np.asarray(dm, dtype=np.float64)
        sc[f"k{k}_{hn}_score_loop"] = np.asarray(s2, dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "scores.npz"), **sc)
with open(os.path.join(OUT, "scores_hparams.json"), "w") as f:
    json.dump({k: {a: float(b) for a, b in v.items()} for k, v in HPS.items()}, f)

# metrics: prob_metrics / optimize_f1(_efficient) / eval_metrics / maximize_metric on a 200-row frame
rs = np.random.RandomState(7)
df, rec, y = make_frame(rs, 200, 5)
met = {f"in_{nm}": arr for nm, arr in rec.items()}
met["in_y"] = y
score = np.asarray(mu.calc_scores_given_hparams_vectorized(df, HPS["fixed"]), dtype=np.float64)
met["score_fixed"] = score
pm = mu.prob_metrics(y, score)
met["AUROC"], met["AUPRC"] = np.float64(pm["AUROC"]), np.float64(pm["AUPRC"])
f1, th = mu.optimize_f1_efficient(y, score, True)
met["f1_eff"], met["thres_eff"] = np.float64(f1), np.float64(th)
f1, th = mu.optimize_f1(y, score, True)
met["f1_grid"], met["thres_grid"] = np.float64(f1), np.float64(th)
f1, th = mu.f1_with_pred_prev_constraint(y, score, y.mean(), True)
met["f1_prev"], met["thres_prev"] = np.float64(f1), np.float64(th)
f1, th = mu.f1_with_local_minima_finder(y, score, True)
met["f1_heur"], met["thres_heur"] = np.float64(f1), np.float64(th)
ev = mu.eval_metrics(y, score, prevalence=y.mean())
ev_keys = sorted(k for k, v in ev.items() if np.isscalar(v))
met["eval_keys"] = np.array(ev_keys)
met["eval_vals"] = np.array([float(ev[k]) for k in ev_keys], dtype=np.float64)
# hyper-parameter search on a reduced grid (full protocol: run_lemon.py:332-337 is 7 056 points)
grid = {"beta": [0, 5, 10], "gamma": [0, 5, 10], "tau_1": [0, 1], "tau_2": [0, 5]}
bx, bv, bt = mu.maximize_metric(df, grid, [[0] * 6, [1] * 6], mu.optimize_f1_efficient, {}, scipy_methods=["Nelder-Mead"])
met["search_best_x"] = np.asarray(bx, dtype=np.float64)
met["search_best_val"] = np.float64(bv)
met["search_best_thres"] = np.float64(bt)
# grid-only optimum (deterministic, no local optimiser): what a batched GPU grid search must reproduce
best = (-1.0, None)
for c in mu.combinations_base(grid):
    g = [c["beta"], c["gamma"], c["tau_1"], c["tau_2"], c["tau_1"], c["tau_2"]]
    v = -mu.optim_func(g, df, mu.optimize_f1_efficient, {})
    if v > best[0]:
        best = (v, g)
met["grid_best_val"] = np.float64(best[0])
met["grid_best_x"] = np.asarray(best[1], dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "metrics.npz"), **met)

# in-tree torch brute-force kNN (dead code lib/metrics/utils.py:198-214): second opinion, untied data
rs = np.random.RandomState(3)
feat = rs.randn(64, 16).astype(np.float32)
dist = mu.cosDistance(torch.from_numpy(feat))
vals, idx = dist.topk(6, dim=1, largest=False, sorted=True)
np.savez_compressed(os.path.join(OUT, "cosdistance_topk.npz"), feat=feat, vals=vals.numpy(), idx=idx.numpy())

print("golden fixtures written to", OUT)
for fn in sorted(os.listdir(OUT)):
    print(f"  {fn:32s} {os.path.getsize(os.path.join(OUT, fn)):8d} B")

# ------------------------------------------------------------------ zero-shot CLIP-logits baseline
# lib/baselines/train_zero_shot_clip_baseline.py:207-224 restated with the reference's OWN pieces: DistanceEvaluator.our_metric
# (lib/metrics/distance_metrics.py:48-73) per image against all class prompts, scipy softmax(1 - dist), entry of the noisy label.
from scipy.special import softmax as sp_softmax
dm = importlib.import_module("lib.metrics.distance_metrics")
rs = np.random.RandomState(11)
zs = {"img": (rs.randn(40, 24) * rs.uniform(0.5, 3, (40, 1))).astype(np.float32),
      "cls": (rs.randn(10, 24) * rs.uniform(0.5, 3, (10, 1))).astype(np.float32), "lab": rs.randint(0, 10, 40).astype(np.int64)}
text_embeds = torch.from_numpy(zs["cls"])
for dist_name in ("cosine", "euclidean", "manhattan"):
    conf = []
    for i in range(40):
        rep = torch.from_numpy(zs["img"][i]).repeat(10, 1)
        ev = dm.DistanceEvaluator(y_true=None, y_pred_proba=None, dist=dist_name, threshold=0.5, y_pred_prob_epochs=None, loss=None,
                                  first_modality_embeddings=text_embeds, second_modality_embeddings=rep)
        conf.append(sp_softmax(1 - ev.our_metric())[zs["lab"][i]])
    zs[f"conf_{dist_name}"] = np.array(conf, dtype=np.float64)
np.savez_compressed(os.path.join(OUT, "zero_shot.npz"), **zs)
print("zero_shot.npz written")
