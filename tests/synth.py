"""Seeded synthetic inputs shared by the tests, bench.py and smoke() (SURVEY 8d "Synthetic inputs")."""
import numpy as np


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def pairflip(rng, y, noise, C):
    """40% pair-flip noise (shape of lib/datasets/utils.py:223-245; own RNG stream)."""
    flip = rng.random(y.shape[0]) < noise
    return np.where(flip, (y + 1) % C, y)


def planted(seed, n_tr, n_q, d, C, noise=0.4, sigma=0.5):
    """S-small / S-cifar generator: text embedding = class prototype of the NOISY label (exact
    duplicates => ties, as on CIFAR), image embedding = normalize(proto[clean] + sigma*randn)."""
    rng = np.random.default_rng(seed)
    proto = unit_rows(rng, C, d)

    def make(n):
        clean = rng.integers(0, C, n)
        noisy = pairflip(rng, clean, noise, C)
        img = proto[clean] + sigma * rng.standard_normal((n, d)).astype(np.float32) / np.sqrt(d).astype(np.float32)
        img = img / np.linalg.norm(img, axis=1, keepdims=True)
        return (np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(proto[noisy]),
                clean.astype(np.int32), noisy.astype(np.int32))

    return {"proto": proto, "train": make(n_tr), "query": make(n_q)}


def second_opinion_inputs():
    """Inputs of tests/golden/knn_second_opinion.npz (tools/make_golden_knn.py): 2 048 x 64 features for the reference's
    cosDistance + topk, and an un-normalised 2 048 x 64 database with 64 queries for its sklearn euclidean metric."""
    rs = np.random.RandomState(31)
    feat = rs.randn(2048, 64).astype(np.float32)
    X = (rs.randn(2048, 64) * rs.uniform(0.5, 2.0, (2048, 1))).astype(np.float32)
    Q = (rs.randn(64, 64) * rs.uniform(0.5, 2.0, (64, 1))).astype(np.float32)
    return feat, X, Q
