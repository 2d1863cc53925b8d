"""Input side of the hot path: label sets, synthetic label noise, splits, caption noise.

Host logic that feeds run_lemon's argument surface; mirrors (by behaviour, pinned by the golden
vectors in tests/golden/) these reference pieces:
  label lists / class_num_dict / CLIP_MEAN, CLIP_STD    lib/datasets/utils.py:27-160
  add_noisy_labels                                      lib/datasets/utils.py:172-193
  noisify_pairflip / noisify_multiclass_symmetric       lib/datasets/utils.py:197-273
  80/10/10 split                                        lib/datasets/utils.py:409-410
  caption noise                                         lib/datasets/noise_captioning.py:4-54
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "data", "label_sets.json")) as _f:
    _META = json.load(_f)

cifar10_labels = np.array(_META["labels"]["cifar10"])
cifar100_labels = np.array(_META["labels"]["cifar100"])
mini_imagenet_labels = np.array(_META["labels"]["mini_imagenet"])
stanford_cars_labels = np.array(_META["labels"]["stanford_cars"])
class_num_dict = dict(_META["class_num_dict"])
CLIP_MEAN = list(_META["CLIP_MEAN"])
CLIP_STD = list(_META["CLIP_STD"])

clf_datasets = ["cifar10", "cifar100", "cifar10_full", "cifar100_full", "mini_imagenet", "stanford_cars"]  # run_lemon.py:32
LABEL_SETS = {
    "cifar10": cifar10_labels, "cifar10_full": cifar10_labels,
    "cifar100": cifar100_labels, "cifar100_full": cifar100_labels,
    "stanford_cars": stanford_cars_labels, "mini_imagenet": mini_imagenet_labels,
}  # run_lemon.py:94-101


# ---------------------------------------------------------------------------- class-label noise
def _transition_noisify(y, P, random_state):
    """Draw a noisy label per sample from row P[y] with the legacy RandomState multinomial stream
    (one draw per sample, in order) -- the stream the reference consumes, so seeds are comparable."""
    y = np.asarray(y)
    assert P.shape[0] == P.shape[1] and y.max() < P.shape[0] and (P >= 0).all()
    rng = np.random.RandomState(random_state)
    out = y.copy()
    for i, cls in enumerate(y):
        out[i] = int(np.flatnonzero(rng.multinomial(1, P[cls], 1)[0] == 1)[0])
    return out


def pairflip_matrix(nb_classes, noise):
    P = np.eye(nb_classes)
    if noise > 0:
        for c in range(nb_classes):
            P[c, c] = 1.0 - noise
            P[c, (c + 1) % nb_classes] = noise
    return P


def symmetric_matrix(nb_classes, noise):
    P = np.full((nb_classes, nb_classes), noise / (nb_classes - 1))
    if noise > 0:
        np.fill_diagonal(P, 1.0 - noise)
    return P


def noisify_pairflip(y_train, noise, random_state=None, nb_classes=10):
    """class c -> c+1 (cyclic) with probability `noise` ("asymmetric")."""
    y_train = np.asarray(y_train)
    if noise <= 0:
        return y_train, 0.0
    y_noisy = _transition_noisify(y_train, pairflip_matrix(nb_classes, noise), random_state)
    actual = float((y_noisy != y_train).mean())
    assert actual > 0.0
    return y_noisy, actual


def noisify_multiclass_symmetric(y_train, noise, random_state=None, nb_classes=10):
    """uniform flip to any other class with total probability `noise` ("symmetric")."""
    y_train = np.asarray(y_train)
    if noise <= 0:
        return y_train, 0.0
    y_noisy = _transition_noisify(y_train, symmetric_matrix(nb_classes, noise), random_state)
    actual = float((y_noisy != y_train).mean())
    assert actual > 0.0
    return y_noisy, actual


def add_noisy_labels(dataset, noise_type, noise_prop, data_seed=1, y_true=None, data_root="./data"):
    """lib/datasets/utils.py:172-193.  `real` needs the CIFAR-N files locally; `cat|noun|random`
    are caption-only and raise NotImplementedError for class datasets exactly like the reference
    (BASELINE's "cifar + cat" is not a runnable combination upstream either)."""
    if noise_type == "real":
        import torch
        fn, key = {"cifar10": ("CIFAR-10_human.pt", "worse_label"),
                   "cifar100": ("CIFAR-100_human.pt", "noisy_label")}[dataset]
        return torch.load(os.path.join(data_root, fn))[key]
    assert y_true is not None
    assert 0 <= noise_prop < 1
    y_true = np.array(y_true)
    if noise_type == "symmetric":
        return noisify_multiclass_symmetric(y_true, noise_prop, data_seed, class_num_dict[dataset])[0]
    if noise_type == "asymmetric":
        return noisify_pairflip(y_true, noise_prop, data_seed, class_num_dict[dataset])[0]
    raise NotImplementedError(noise_type)


def split_80_10_10(n, data_seed):
    """train/val/test index arrays, lib/datasets/utils.py:409-410 (two sklearn shuffles)."""
    from sklearn.model_selection import train_test_split
    tr, rest = train_test_split(np.arange(n), test_size=0.2, random_state=data_seed)
    va, te = train_test_split(rest, test_size=0.5, random_state=data_seed)
    return tr, va, te


def split_80_20(n, data_seed):
    """train/val index arrays of the *_full datasets, lib/datasets/utils.py:389 (one sklearn shuffle; the test set is the
    dataset's own test split)."""
    from sklearn.model_selection import train_test_split
    return train_test_split(np.arange(n), test_size=0.2, random_state=data_seed)


# ---------------------------------------------------------------------------- caption noise
def random_noise_dict(num_items, frac_noise=0.3, seed=42):
    """{row -> row whose caption it receives}; lib/datasets/noise_captioning.py:35-42."""
    rng = np.random.default_rng(seed)
    chosen = rng.choice(np.arange(num_items), int(frac_noise * num_items), replace=False)
    out = {}
    everything = np.arange(num_items)
    for i in chosen:
        out[i] = rng.choice(np.delete(everything, i), 1)[0]
    return out


def calc_noise_by_integer_matching(cat_labels, frac_noise=0.3, seed=42):
    """Swap captions between samples sharing a category / noun id; noise_captioning.py:4-33."""
    n = len(cat_labels)
    sets = [set(row) for row in cat_labels]
    top = max(max(row) for row in cat_labels if len(row) > 0) + 1
    members = {c: [i for i, s in enumerate(sets) if c in s] for c in range(top)}
    rng = np.random.default_rng(seed)
    eligible = [i for i in range(n) if len(cat_labels[i]) > 0]
    chosen = rng.choice(eligible, int(frac_noise * n), replace=False)
    out = {}
    for i in chosen:
        c = rng.choice(cat_labels[i])
        pool = np.setdiff1d(members[c], [i])
        if len(pool) > 0:
            out[i] = rng.choice(pool, 1)[0]
    return out


def noise_given_dict(meta, d):
    """Apply a {source position -> target position} caption swap to a DataFrame with a `sentence`
    column; adds gold_sentence / is_mislabel.  noise_captioning.py:44-54."""
    out = meta.copy()
    out["gold_sentence"] = out["sentence"]
    src = meta.index[list(d.keys())]
    dst = meta.index[list(d.values())]
    out.loc[src, "sentence"] = meta.loc[dst, "sentence"].values
    out["is_mislabel"] = out["sentence"] != out["gold_sentence"]
    return out
