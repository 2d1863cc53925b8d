"""Second opinions on the flat search from the REFERENCE's own in-tree brute-force code, at a size where the scan kernels
run several tiles and a deep k (N = 2 048, d = 64, k = 51):

  * inner product: `cosDistance(features).topk(51, largest=False, sorted=True)` (lib/metrics/utils.py:198-214, the
    torch N x N matrix the reference's count_knn_distribution searches) -- stored for every 4th row;
  * L2: `DistanceEvaluator(dist='euclidean').our_metric()` (lib/metrics/distance_metrics.py:48-73: the diagonal of sklearn's
    full pairwise euclidean_distances) evaluated once per query against all database rows, then a stable argsort -- the
    reference's own (non-squared) euclidean arithmetic against IndexFlatL2's squared distances.

Inputs are regenerated from seeds by the tests (tests/test_oracle_golden.py, tests/test_gpu_parity.py); stored: outputs only.
Neither pins faiss (absent); both are independent float32 implementations of the same exact search.
Run:  python tools/make_golden_knn.py   -> tests/golden/knn_second_opinion.npz
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
from tests.synth import second_opinion_inputs      # noqa: E402  (seeded recipe shared with the tests)

for m in ("torchvision", "torchvision.transforms", "torchvision.datasets", "lib.datasets.clustering"):
    sys.modules[m] = MagicMock()
netcal, netcal_m = types.ModuleType("netcal"), types.ModuleType("netcal.metrics")
netcal_m.ECE = type("ECE", (), {"__init__": lambda self, *a, **k: None, "measure": lambda self, a, b: 0.0})
netcal.metrics = netcal_m
sys.modules.update({"netcal": netcal, "netcal.metrics": netcal_m})
mu = importlib.import_module("lib.metrics.utils")
dm = importlib.import_module("lib.metrics.distance_metrics")

K = 51
feat, X, Q = second_opinion_inputs()
out = {}
dist = mu.cosDistance(torch.from_numpy(feat))
vals, idx = dist.topk(K, dim=1, largest=False, sorted=True)
out["ip_vals"], out["ip_idx"] = vals.numpy()[::4].astype(np.float32), idx.numpy()[::4].astype(np.int16)
gap = np.diff(vals.numpy()[::4], axis=1).min()
print("cosDistance top-51: smallest gap between consecutive distances", gap)

Xt = torch.from_numpy(X)
l2_d, l2_i = [], []
for q in Q:
    ev = dm.DistanceEvaluator(y_true=None, y_pred_proba=None, dist="euclidean", threshold=0.5, y_pred_prob_epochs=None, loss=None,
                              first_modality_embeddings=Xt, second_modality_embeddings=torch.from_numpy(q).repeat(len(X), 1))
    dq = np.asarray(ev.our_metric())
    order = np.argsort(dq, kind="stable")[:K]
    l2_i.append(order.astype(np.int16)); l2_d.append(dq[order].astype(np.float32))
out["l2_dist"], out["l2_idx"] = np.stack(l2_d), np.stack(l2_i)
print("euclidean top-51: smallest gap", np.diff(out["l2_dist"], axis=1).min())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "knn_second_opinion.npz"), **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(ROOT, "tests", "golden", "knn_second_opinion.npz")), "B")
