"""Loader for tests/golden/loop_*.npz: inputs and outputs of the REFERENCE's own run_lemon.py loop executed by
tools/make_golden_loop.py (runpy + stand-ins for faiss / the CLIP weights / the datasets; see that file's
header for what is real and what is a stand-in: the fixtures pin the loop around the search, not faiss)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REC = ("dists_n", "D_n", "dists_tr_n", "dists_m", "D_m", "dists_tr_m")


def case_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "loop_*.npz")))


class LoopCase:
    def __init__(self, name):
        self.name = name
        self.fx = np.load(os.path.join(GOLDEN, f"loop_{name}.npz"), allow_pickle=False)
        self.argv = json.loads(str(self.fx["argv"]))
        a = self.argv

        def opt(flag, default, conv=str):
            return conv(a[a.index(flag) + 1]) if flag in a else default

        self.k = opt("--knn_k", 5, int)
        self.metric = opt("--dist_type", "cosine")
        self.discrete = "--use_discrete_for_text" in a
        self.normalize_d1 = "--normalize_d1" in a
        self.dataset = opt("--dataset", "cifar100")
        self.ablation = opt("--ablation", "none")         # 'd1' zeroes the d_1 column of the reference's frame (run_lemon.py:316-317)
        self.is_caption = bool(self.fx["is_caption"])
        self.ssets = [str(s) for s in self.fx["ssets"]]
        self.n_train = int(self.fx["n_train"])
        self.sel = self.fx["train_indices_in_compr"]
        self.agg = json.loads(str(self.fx["agg_results"])) if "agg_results" in self.fx else None

    # ---- inputs of the scoring loop exactly as the reference handed them to the index
    def db(self):
        return self.fx["db_img"], self.fx["db_txt"]

    def queries(self, s):
        return self.fx[f"{s}_q_img"], self.fx[f"{s}_q_txt"]

    def in_db(self, s):
        """`sample_idx in train_indices_in_compr` (run_lemon.py:258): only meaningful for the train split."""
        n = len(self.fx[f"{s}_d_1"])
        return np.isin(np.arange(n), self.sel).astype(np.uint8) if s == "train" else None

    def prompts(self, s):
        """noisy_text_labels_prompts of a split (run_lemon.py:210,213): prefix + label for class datasets, the caption
        otherwise."""
        t = self.fx[f"{s}_noisy_text"]
        if self.is_caption:
            return [str(c) for c in t]
        prefix = str(self.fx["prefix"])
        return [prefix + str(c) for c in t]

    def label_ids(self, s):
        """int ids standing for the prompt STRINGS the discrete text metric compares (run_lemon.py:266-267):
        (ids of the DB rows, ids of the split's queries) under one string -> id dictionary."""
        vocab = {}
        tr = np.array([vocab.setdefault(str(c), len(vocab)) for c in self.fx["db_text_labels"]], np.int32)
        q = np.array([vocab.setdefault(c, len(vocab)) for c in self.prompts(s)], np.int32)
        return tr, q

    def expected(self, s):
        out = {c: self.fx[f"{s}_{c}"] for c in REC}
        out["d_1"] = self.fx[f"{s}_d_1"]
        return out

    def expected_I(self, s):
        """I_n / I_m after the reference's self-exclusion rule applied to the raw (k+1) search results."""
        res = {}
        for side, key in (("img", "I_n"), ("txt", "I_m")):
            I = self.fx[f"{s}_search_I_{side}"]
            if s == "train":
                m = self.in_db(s).astype(bool)[:, None]
                I = np.where(m, I[:, 1:], I[:, :-1])
            res[key] = I
        return res


def assert_records_match(got, case, s, float_tol=1e-6):
    """Index sets bit-exact; D_n / D_m (search output, chain arithmetic on both sides) bit-exact; the quantities the
    reference computes with torch reductions (d_1, dists_*, dists_tr) within float_tol (north_star: scores 1e-4)."""
    exp, expI = case.expected(s), case.expected_I(s)
    if "d1" in case.ablation:
        got = dict(got, d_1=np.zeros_like(np.asarray(got["d_1"])))
    for key in ("I_n", "I_m"):
        if key in got:
            assert np.array_equal(np.asarray(got[key]), expI[key]), f"{case.name}/{s}/{key}"
    for key in ("D_n", "D_m"):
        assert np.array_equal(np.asarray(got[key]), exp[key]), f"{case.name}/{s}/{key}"
    if case.discrete:
        assert np.array_equal(np.asarray(got["dists_n"]), exp["dists_n"]), f"{case.name}/{s}/dists_n (discrete)"
    for key in ("d_1", "dists_n", "dists_m", "dists_tr_n", "dists_tr_m"):
        d = np.abs(np.asarray(got[key], np.float64) - exp[key]).max()
        assert d <= float_tol, f"{case.name}/{s}/{key}: max abs diff {d}"
