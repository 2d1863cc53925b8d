"""Pins the oracle's per-sample loop to the REFERENCE's own loop: executes /root/reference/run_lemon.py
(module-level script, `runpy`) in the build container on small planted datasets and writes inputs +
the reference's outputs to tests/golden/loop_*.npz.

What runs for real (reference code, unmodified, from /root/reference):
  run_lemon.py:1-436 (argument parsing, DB subset draw :121-127, DB embedding loop :137-176, class
  prompt embeddings :180-190, the per-sample scoring loop :198-307, the hyper-parameter search and
  eval :319-427), lib/datasets/utils.py get_dataset / add_noisy_labels / get_captioning_dataset,
  lib/datasets/dataloader.py NoisyCombinedDataset / CaptioningDataset, lib/datasets/noise_captioning.py,
  lib/utils/utils.py normalize_vectors, lib/metrics/utils.py (maximize_metric, eval_metrics, ...),
  torch DataLoader + default collate, scipy softmax, sklearn.

What is a stand-in (modules absent from this image, or weights/data that need a network):
  faiss                  numpy exact brute force.  Scores are the float32 fma chain of DESIGN.md section 2
                         (emulated exactly: float64 TwoSum + round-to-odd, then one rounding to float32);
                         ties go to the lower DB index.  THIS STAND-IN PINS THE LOOP AROUND THE SEARCH
                         (self-exclusion, sign quirk, discrete text metric, normalize_d1, DB subset,
                         record schema), NOT faiss's arithmetic or tie order, which stay unpinned.
  lib.models.utils       a planted "CLIP": encode_image returns the planted vector carried as the pixel
                         tensor, encode_text looks the prompt's vector up; tokenizer = prompt -> row id.
  torchvision            MagicMock + an in-memory CIFAR10/CIFAR100 class (targets + planted vectors);
                         generic_transform = identity.
  netcal, lib.datasets.clustering   unused on this path.
  DataLoader             num_workers forced to 0 (same batches, no fork).
  caption data           a synthetic multimodal_mislabel_split.pkl under a temp PATHS['mscoco'].
  stanford_cars          a synthetic multimodal_mislabel_split.csv + REAL image files (lossless PNG, a few pixels each) under
                         a temp PATHS['stanford_cars']: get_large_scale_dataset (lib/datasets/utils.py:325-347) and
                         LargeScaleDataset incl. its PIL Image.open / `real_label = noisy - 1` (lib/datasets/dataloader.py:
                         113-133) run for real; generic_transform = "the planted vector whose float32 bytes ARE the pixels".

Outputs are DATA (inputs + expected outputs); no reference source text is stored.
Run:  python tools/make_golden_loop.py [--only NAME]
"""
import argparse
import contextlib
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
FLT_MAX = np.finfo(np.float32).max


# ------------------------------------------------------------------------------ exact float32 fma chain in numpy
def fma32(a, b, c):
    """fl32(a*b + c) with ONE rounding, for float32 arrays.  a*b is exact in float64; the float64 sum is
    turned into round-to-odd with the TwoSum error term, so the final cast to float32 cannot double-round."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)
    fix = (e != 0) & ((s.view(np.int64) & 1) == 0)
    s = np.where(fix, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def chain_ip(Q, X):
    """[nq, n] matrix of dot(q, x) = fma chain in ascending k starting from +0."""
    acc = np.zeros((Q.shape[0], X.shape[0]), np.float32)
    for kk in range(Q.shape[1]):
        acc = fma32(Q[:, kk, None], X[None, :, kk], acc)
    return acc


def chain_self(X):
    acc = np.zeros(X.shape[0], np.float32)
    for kk in range(X.shape[1]):
        acc = fma32(X[:, kk], X[:, kk], acc)
    return acc


class _FakeIndex:
    """faiss.IndexFlatIP / IndexFlatL2 surface used by run_lemon.py:167-176,235-236."""
    log = None      # the active capture dict (set per run)

    def __init__(self, d):
        self.d, self.x = int(d), np.zeros((0, int(d)), np.float32)
        self.name = None

    @property
    def ntotal(self):
        return self.x.shape[0]

    def add(self, x):
        assert isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and x.shape[1] == self.d
        self.x = np.concatenate([self.x, x.copy()])
        _FakeIndex.log["adds"].append((self, x.copy()))

    def search(self, q, k):
        assert isinstance(q, np.ndarray) and q.dtype == np.float32 and q.shape[1] == self.d
        k = int(k)
        s = chain_ip(q, self.x)
        if self.metric == "l2":
            s = np.maximum(np.float32(0), fma32(np.float32(-2.0) * np.ones_like(s), s,
                                                (chain_self(q)[:, None] + chain_self(self.x)[None, :]).astype(np.float32)))
            key = s
        else:
            key = -s
        n = self.x.shape[0]
        idx = np.broadcast_to(np.arange(n), key.shape)
        order = np.lexsort((idx, key), axis=1)[:, :k]          # (key asc, index asc)
        D = np.take_along_axis(s, order, 1).astype(np.float32)
        I = order.astype(np.int64)
        if k > n:
            pad = k - n
            D = np.concatenate([D, np.full((q.shape[0], pad), FLT_MAX if self.metric == "l2" else -FLT_MAX, np.float32)], 1)
            I = np.concatenate([I, np.full((q.shape[0], pad), -1, np.int64)], 1)
        _FakeIndex.log["searches"].append((self, q.copy(), k, D.copy(), I.copy()))
        return D, I


class IndexFlatIP(_FakeIndex):
    metric = "ip"


class IndexFlatL2(_FakeIndex):
    metric = "l2"


# ------------------------------------------------------------------------------ planted data
def modality_gap(rs, d):
    """Two orthogonal offsets, one per modality, 1.7x the prototype norm: image-text cosine of a matching pair
    ~0.25 and within-modality cosines 0.75-0.95, the geometry CLIP embeddings have.  (Without the gap the
    reference's own LBFGS polish, lib/metrics/utils.py:121-149, overflows its SoftMargin proxy from the start
    point [10]*6 and the reference dies in fminbound -- see tests/test_host_logic.py for that case.)"""
    u = np.linalg.qr(rs.randn(d, 2))[0].T.astype(np.float32)
    g = 1.7 * np.sqrt(d)
    return g * u[0], g * u[1]


def planted_class_data(seed, n, C, d):
    rs = np.random.RandomState(seed)
    proto = rs.randn(C, d).astype(np.float32)
    g_img, g_txt = modality_gap(rs, d)
    y = rs.randint(0, C, n).astype(np.int64)
    scale = rs.uniform(0.5, 3.0, (n, 1)).astype(np.float32)
    img = ((proto[y] + 0.5 * rs.randn(n, d) + g_img) * scale).astype(np.float32)
    txt_table = ((proto + 0.3 * rs.randn(C, d) + g_txt) * rs.uniform(0.5, 3.0, (C, 1))).astype(np.float32)
    return img, y, txt_table


def planted_caption_frame(seed, n, d, n_cat=12):
    """Synthetic multimodal_mislabel_split.pkl: index = cocoid-like ints, splits incl. 'restval' (dropped by the
    reference's no-op remap), a few duplicated sentences, category / noun id lists."""
    rs = np.random.RandomState(seed)
    proto = rs.randn(n_cat, d).astype(np.float32)
    g_img, g_txt = modality_gap(rs, d)
    cats = [sorted(set(rs.randint(0, n_cat, rs.randint(1, 3)).tolist())) for _ in range(n)]
    for j in range(0, n, 37):
        cats[j] = []                                      # rows without categories cannot be matched
    nouns = [sorted(set(rs.randint(0, 30, rs.randint(1, 4)).tolist())) for _ in range(n)]
    first = np.array([c[0] if c else rs.randint(0, n_cat) for c in cats])
    sent = [f"a synthetic caption number {i} about category {first[i]}" for i in range(n)]
    for j in range(5, n, 41):
        sent[j] = sent[j - 5]                             # exact duplicate captions exist in COCO as well
    split = np.array(["train"] * n, dtype=object)
    perm = rs.permutation(n)
    n_val = n_test = n // 10
    split[perm[:n_val]] = "val"
    split[perm[n_val:n_val + n_test]] = "test"
    split[perm[n_val + n_test:n_val + n_test + n // 10]] = "restval"
    cocoid = 100000 + rs.permutation(n * 3)[:n]
    img = ((proto[first] + 0.5 * rs.randn(n, d) + g_img) * rs.uniform(0.5, 3.0, (n, 1))).astype(np.float32)
    uniq = sorted(set(sent))
    sid = {s: i for i, s in enumerate(uniq)}
    cat_of = {}
    for s, c in zip(sent, first):
        cat_of.setdefault(s, c)
    txt_table = np.stack([proto[cat_of[s]] for s in uniq]).astype(np.float32)
    txt_table = ((txt_table + 0.3 * rs.randn(len(uniq), d) + g_txt) * rs.uniform(0.5, 3.0, (len(uniq), 1))).astype(np.float32)
    df = pd.DataFrame({"split": split, "filepath": "synthetic", "filename": [f"{c}.jpg" for c in cocoid],
                       "sentence": sent, "cat_labels": cats, "nouns_int": nouns}, index=cocoid)
    return df, img, uniq, sid, txt_table


def vec_to_png(vec, path):
    """float32 vector -> lossless RGB PNG whose raw pixel bytes start with the vector's bytes (d = 32: 128 of 132 bytes)."""
    from PIL import Image
    raw = np.asarray(vec, np.float32).tobytes()
    w = -(-len(raw) // 12)
    buf = np.zeros(4 * w * 3, np.uint8)
    buf[:len(raw)] = np.frombuffer(raw, np.uint8)
    Image.fromarray(buf.reshape(4, w, 3), "RGB").save(path, format="PNG")


def png_to_vec(img, d):
    return np.frombuffer(np.asarray(img, np.uint8).tobytes()[:4 * d], np.float32).copy()


def planted_large_scale(seed, n, C, d):
    """multimodal_mislabel_split.csv of a 'real noise' dataset: filename, label (the possibly wrong web label), is_clean."""
    rs = np.random.RandomState(seed)
    proto = rs.randn(C, d).astype(np.float32)
    g_img, g_txt = modality_gap(rs, d)
    true = rs.randint(0, C, n).astype(np.int64)
    is_clean = rs.rand(n) > 0.3
    label = np.where(is_clean, true, (true + rs.randint(1, C, n)) % C).astype(np.int64)
    scale = rs.uniform(0.5, 3.0, (n, 1)).astype(np.float32)
    img = ((proto[true] + 0.5 * rs.randn(n, d) + g_img) * scale).astype(np.float32)
    txt_table = ((proto + 0.3 * rs.randn(C, d) + g_txt) * rs.uniform(0.5, 3.0, (C, 1))).astype(np.float32)
    files = np.array([f"car_images/{i:05d}.png" for i in range(n)])
    return img, label, is_clean, files, txt_table


# ------------------------------------------------------------------------------ stand-in modules
class PlantedCLIP(torch.nn.Module):
    def __init__(self, txt_table, img_log=None):
        super().__init__()
        self.register_buffer("table", torch.from_numpy(txt_table))
        self.context_length = 8
        self.img_log = img_log

    def encode_image(self, pixel_values=None):
        if self.img_log is not None:
            self.img_log.append(pixel_values.float().numpy().copy())
        return pixel_values.float().clone()

    def encode_text(self, input_ids=None, attention_mask=None):
        return self.table[input_ids[:, 0]].clone()


class PlantedTokenizer:
    """HF-callable for huggingface_clip (dict of lists), LongTensor-callable for the other branches."""

    def __init__(self, prompt_ids, hf):
        self.prompt_ids, self.hf = prompt_ids, hf

    def __call__(self, texts, padding=None, truncation=None):
        ids = [[self.prompt_ids[t], 0] for t in texts]
        if self.hf:
            assert padding == "max_length" and truncation is True
            return {"input_ids": ids, "attention_mask": [[1, 0] for _ in ids]}
        return torch.tensor(ids, dtype=torch.long)


def install_stubs(state):
    """state: dict the stand-ins read their planted data from (set per run)."""
    sys.path.insert(0, REF)
    from transformers import AutoTokenizer  # noqa: F401  (imported BEFORE the torchvision mock: transformers probes for it)
    faiss = types.ModuleType("faiss")
    faiss.IndexFlatIP, faiss.IndexFlatL2 = IndexFlatIP, IndexFlatL2
    sys.modules["faiss"] = faiss

    tv, tvt, tvd, tvm = MagicMock(), MagicMock(), MagicMock(), MagicMock()
    tv.transforms, tv.datasets, tv.models = tvt, tvd, tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.datasets": tvd,
                        "torchvision.models": tvm, "lib.datasets.clustering": MagicMock()})

    class FakeCIFAR:
        def __init__(self, root=None, train=True, download=False, transform=None):
            if train:
                self.x, self.targets = state["img_all"], [int(v) for v in state["y_all"]]
            else:                       # the dataset's own test split (cifar10_full / cifar100_full, utils.py:379-381)
                self.x, self.targets = state["img_test"], [int(v) for v in state["y_test"]]

        def __len__(self):
            return len(self.targets)

        def __getitem__(self, i):
            return torch.from_numpy(self.x[i]), self.targets[i]

    tvd.CIFAR10 = tvd.CIFAR100 = FakeCIFAR

    netcal, netcal_m = types.ModuleType("netcal"), types.ModuleType("netcal.metrics")
    netcal_m.ECE = type("ECE", (), {"__init__": lambda self, *a, **k: None, "measure": lambda self, a, b: 0.0})
    netcal.metrics = netcal_m
    sys.modules.update({"netcal": netcal, "netcal.metrics": netcal_m})

    mu = types.ModuleType("lib.models.utils")

    def algorithm_class_from_scratch(name, text_base_name, img_base, return_tokenizer=False):
        model = PlantedCLIP(state["txt_table"], state.get("img_log"))
        tok = PlantedTokenizer(state["prompt_ids"], hf=(name == "huggingface_clip"))
        return (model, tok) if return_tokenizer else model

    mu.algorithm_class_from_scratch = algorithm_class_from_scratch
    mu.get_img_base = lambda *a, **k: None
    import lib.models  # noqa: F401  (real, empty package)
    sys.modules["lib.models.utils"] = mu

    import torch.utils.data as tud
    real_loader = tud.DataLoader

    class Loader0(real_loader):
        def __init__(self, *a, **k):
            k["num_workers"] = 0
            super().__init__(*a, **k)

    tud.DataLoader = Loader0

    import lib.datasets.utils as dsu
    import lib.datasets.dataloader as dl
    ident = lambda x: x
    dsu.generic_transform = dsu.transform = ident
    dl.CaptioningDataset.get_image = lambda self, path: torch.from_numpy(state["img_by_file"][os.path.basename(str(path))])
    return dsu


# ------------------------------------------------------------------------------ one reference run
REC_COLS = ("dists_n", "D_n", "dists_tr_n", "dists_m", "D_m", "dists_tr_m")


def run_case(name, cfg, dsu, state, workdir):
    d = cfg["d"]
    argv = ["--output_dir", os.path.join(workdir, name)] + cfg["argv"]
    is_caption = cfg["dataset"] in ("mscoco",)
    fx = {"argv": np.array(json.dumps(cfg["argv"])), "is_caption": np.array(is_caption), "d": np.array(d)}
    if is_caption:
        df, img, uniq, sid, txt_table = planted_caption_frame(cfg["seed"], cfg["n"], d)
        root = os.path.join(workdir, name + "_coco")
        os.makedirs(root, exist_ok=True)
        df.to_pickle(os.path.join(root, "multimodal_mislabel_split.pkl"))
        dsu.PATHS["mscoco"] = root
        state.update(txt_table=txt_table, prompt_ids=sid,
                     img_by_file={fn: img[i] for i, fn in enumerate(df["filename"])})
        fx.update(frame_index=df.index.values.astype(np.int64), frame_split=df["split"].values.astype(str),
                  frame_filename=df["filename"].values.astype(str), frame_sentence=df["sentence"].values.astype(str),
                  frame_cat_flat=np.array([c for row in df["cat_labels"] for c in row], np.int64),
                  frame_cat_len=np.array([len(r) for r in df["cat_labels"]], np.int64),
                  frame_noun_flat=np.array([c for row in df["nouns_int"] for c in row], np.int64),
                  frame_noun_len=np.array([len(r) for r in df["nouns_int"]], np.int64),
                  img_all=img, captions=np.array(uniq), txt_table=txt_table)
    elif cfg["dataset"] in ("stanford_cars", "mini_imagenet"):
        labels = np.array(getattr(dsu, cfg["dataset"] + "_labels"))
        img, label, is_clean, files, txt_table = planted_large_scale(cfg["seed"], cfg["n"], len(labels), d)
        root = os.path.join(workdir, name + "_" + cfg["dataset"])
        os.makedirs(os.path.join(root, "car_images"), exist_ok=True)
        for v, f in zip(img, files):
            vec_to_png(v, os.path.join(root, f))
        pd.DataFrame({"filename": files, "label": label, "is_clean": is_clean}).to_csv(
            os.path.join(root, "multimodal_mislabel_split.csv"), index=False)
        dsu.PATHS[cfg["dataset"]] = root
        dsu.generic_transform = lambda im: torch.from_numpy(png_to_vec(im, d))      # (restored below)
        prefix = "A photo of a "
        state.update(txt_table=txt_table, prompt_ids={prefix + l: i for i, l in enumerate(labels)})
        fx.update(img_all=img, csv_filename=files.astype(str), csv_label=label, csv_is_clean=is_clean.astype(np.int64),
                  txt_table=txt_table, prefix=np.array(prefix))
    else:
        base = cfg["dataset"].replace("_full", "")
        C = {"cifar10": 10, "cifar100": 100}[base]
        img, y, txt_table = planted_class_data(cfg["seed"], cfg["n"] + cfg.get("n_test", 0), C, d)
        labels = np.array(getattr(dsu, base + "_labels"))
        prefix = cfg.get("prefix", "A photo of a ")
        n_tr = cfg["n"]
        state.update(img_all=img[:n_tr], y_all=y[:n_tr], img_test=img[n_tr:], y_test=y[n_tr:], txt_table=txt_table,
                     prompt_ids={prefix + l: i for i, l in enumerate(labels)})
        fx.update(img_all=img[:n_tr], y_all=y[:n_tr], txt_table=txt_table, prefix=np.array(prefix))
        if cfg.get("n_test"):
            fx.update(img_test=img[n_tr:], y_test=y[n_tr:])

    log = {"adds": [], "searches": []}
    _FakeIndex.log = log
    old = (sys.argv, sys.stdout, sys.stderr, os.getcwd())
    sys.argv = [os.path.join(REF, "run_lemon.py")] + argv
    os.chdir(workdir)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            g = runpy.run_path(os.path.join(REF, "run_lemon.py"), run_name="__main__")
    finally:
        so, se = sys.stdout, sys.stderr
        sys.argv = old[0]
        sys.stdout, sys.stderr = old[1], old[2]
        for t in (so, se):                                   # the reference's Tee objects keep files open
            f = getattr(t, "file", None)
            if f is not None and f is not old[1] and f is not old[2]:
                with contextlib.suppress(Exception):
                    f.close()
        os.chdir(old[3])
        dsu.generic_transform = dsu.transform

    df = g["df"]
    fx["train_indices_in_compr"] = np.asarray(g["train_indices_in_compr"], np.int64)
    fx["n_train"] = np.array(len(g["train_set"]))
    # DB exactly as handed to faiss (run_lemon.py:175-176: txt first, then img)
    (ix_txt, db_txt), (ix_img, db_img) = log["adds"]
    assert ix_txt is g["index_txt"] and ix_img is g["index_img"]
    fx["db_txt"], fx["db_img"] = db_txt, db_img
    fx["dists_tr"] = np.asarray(g["dists_tr"], np.float32)
    fx["db_text_labels"] = np.asarray(g["tr_text_labels"]).astype(str)      # prompt strings of the DB rows (:146,177)
    if "text_embeds_dataset_labels" in g:
        fx["cls_txt"] = g["text_embeds_dataset_labels"].numpy()
    # queries per split, in batch order (:235 img, :236 txt)
    per = {}
    for ix, q, k, D, I in log["searches"]:
        per.setdefault("img" if ix is ix_img else "txt", []).append((q, k, D, I))
    pos = {"img": 0, "txt": 0}
    for sname in df.sset.unique():
        sub = df.loc[df.sset == sname]
        n = len(sub)
        for side in ("img", "txt"):
            qs, Ds, Is, got = [], [], [], 0
            while got < n:
                q, k, D, I = per[side][pos[side]]
                pos[side] += 1
                qs.append(q); Ds.append(D); Is.append(I); got += len(q)
            assert got == n
            fx[f"{sname}_q_{side}"] = np.concatenate(qs)
            fx[f"{sname}_search_D_{side}"] = np.concatenate(Ds)
            fx[f"{sname}_search_I_{side}"] = np.concatenate(Is)
        assert np.array_equal(sub["idx"].values, np.arange(n))
        fx[f"{sname}_d_1"] = sub["d_1"].values.astype(np.float64)
        for c in REC_COLS:
            fx[f"{sname}_{c}"] = np.stack(sub[c].values).astype(np.float32)
        fx[f"{sname}_is_mislabel"] = sub["is_mislabel"].values.astype(np.int64)
        fx[f"{sname}_noisy_text"] = sub["noisy_label_text"].values.astype(str)
        fx[f"{sname}_clean_text"] = sub["actual_label_text"].values.astype(str)
        if not is_caption:      # (LargeScaleDataset yields numpy ints: run_lemon.py:291-307 stores them as they come)
            fx[f"{sname}_noisy"] = np.array([int(v) for v in sub["noisy_label"]], np.int64)
            fx[f"{sname}_clean"] = np.array([int(v) for v in sub["actual_label"]], np.int64)
    assert pos["img"] == len(per["img"]) and pos["txt"] == len(per["txt"])
    fx["ssets"] = np.array(list(df.sset.unique()))
    res = g["res"]
    if "agg_results" in res:
        sel = res["agg_results"]["know_val_labels"]
        flat = {}
        for key, v in sel.items():
            if isinstance(v, dict):
                flat[key] = {a: float(b) for a, b in v.items() if np.isscalar(b)}
            else:
                flat[key] = float(v)
        fx["agg_results"] = np.array(json.dumps(flat))
        fx["pred_score"] = df["know_val_labels_pred_score"].values.astype(np.float64)
        fx["pred_d_n"] = np.asarray(df["know_val_labels_d_n"].values, np.float64)
        fx["pred_d_m"] = np.asarray(df["know_val_labels_d_m"].values, np.float64)
    fx["out_files"] = np.array(sorted(os.listdir(os.path.join(workdir, name))))
    fx["throughput_line"] = np.array([l for l in buf.getvalue().splitlines() if l.startswith("Finished")][0].split(" in ")[0])
    np.savez_compressed(os.path.join(OUT, f"loop_{name}.npz"), **fx)
    return fx


def run_disc_case(name, cfg, dsu, state, workdir):
    """lib/baselines/discrepancy_baseline.py (module-level script, :1-273) under the same stand-ins: stores the planted inputs,
    the normalised DB / query embeddings it used and its pred_score column + agg_results AUROC."""
    d = cfg["d"]
    argv = ["--output_dir", os.path.join(workdir, name)] + cfg["argv"]
    C = {"cifar10": 10, "cifar100": 100}[cfg["dataset"]]
    img, y, txt_table = planted_class_data(cfg["seed"], cfg["n"], C, d)
    labels = np.array(getattr(dsu, cfg["dataset"] + "_labels"))
    state.update(img_all=img, y_all=y, txt_table=txt_table, prompt_ids={"A photo of a " + l: i for i, l in enumerate(labels)},
                 img_log=[])
    fx = {"argv": np.array(json.dumps(cfg["argv"])), "is_caption": np.array(False), "d": np.array(d), "img_all": img, "y_all": y,
          "txt_table": txt_table, "prefix": np.array("A photo of a ")}
    log = {"adds": [], "searches": []}
    _FakeIndex.log = log
    old = (sys.argv, sys.stdout, sys.stderr, os.getcwd())
    script = os.path.join(REF, "lib", "baselines", "discrepancy_baseline.py")
    sys.argv = [script] + argv
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            g = runpy.run_path(script, run_name="__main__")
    finally:
        so, se = sys.stdout, sys.stderr
        sys.argv = old[0]
        sys.stdout, sys.stderr = old[1], old[2]
        for t in (so, se):
            f = getattr(t, "file", None)
            if f is not None and f is not old[1] and f is not old[2]:
                with contextlib.suppress(Exception):
                    f.close()
        os.chdir(old[3])
        img_log = state.pop("img_log")
    df = g["df"]
    (ix_txt, db_txt), (ix_img, db_img) = log["adds"]
    fx["db_txt"], fx["db_img"] = db_txt, db_img
    fx["train_indices_in_compr"] = np.asarray(g["train_indices_in_compr"], np.int64)
    n_db_batches = -(-len(db_img) // g["bs"])
    raw_q = np.concatenate(img_log[n_db_batches:])                     # image "embeddings" of the scored splits, in order
    first = 1 if "dis" in g["args"].method else 0                      # 'dis_*' searches the DB against itself first (:165-166)
    q_txt = np.concatenate([q for ix, q, k_, D, I in log["searches"][first:]])
    lo = 0
    for sname in df.sset.unique():
        sub = df.loc[df.sset == sname]
        n = len(sub)
        fx[f"{sname}_q_img_raw"] = raw_q[lo:lo + n]
        fx[f"{sname}_q_txt"] = q_txt[lo:lo + n]
        fx[f"{sname}_pred_score"] = sub["pred_score"].values.astype(np.float64)
        fx[f"{sname}_is_mislabel"] = sub["is_mislabel"].values.astype(np.int64)
        fx[f"{sname}_noisy"] = np.array([int(v) for v in sub["noisy_label"]], np.int64)
        lo += n
    assert lo == len(raw_q) == len(q_txt)
    fx["ssets"] = np.array(list(df.sset.unique()))
    fx["auroc"] = np.array(json.dumps({s: float(g["selection_results"][s]["AUROC"]) for s in df.sset.unique()}))
    fx["out_files"] = np.array(sorted(os.listdir(os.path.join(workdir, name))))
    np.savez_compressed(os.path.join(OUT, f"disc_{name}.npz"), **fx)
    return fx


DISC_CASES = {
    m: dict(dataset="cifar10", n=600, d=32, seed=30 + i,
            argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "4", "--method", m] +
                 (["--compr_dataset_size_limit", "300"] if m == "div_y" else []) + (["--skip_train"] if m == "dis_y" else []))
    for i, m in enumerate(("dis_x", "dis_y", "div_x", "div_y"))
}

SKIP = ["--skip_hparam_optim"]
CASES = {
    # class datasets: text rows are exact duplicates per class (all ties on the text side, as on CIFAR)
    "c10_cos_k5_full": dict(dataset="cifar10", n=1000, d=32, seed=11,
                            argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5"]),
    "c10_cos_k5_discrete": dict(dataset="cifar10", n=1000, d=32, seed=12,
                                argv=["--dataset", "cifar10", "--noise_type", "symmetric", "--knn_k", "5",
                                      "--use_discrete_for_text"] + SKIP),
    "c100_cos_k50_nd1": dict(dataset="cifar100", n=1000, d=32, seed=13,
                             argv=["--dataset", "cifar100", "--noise_type", "asymmetric", "--knn_k", "50",
                                   "--normalize_d1"] + SKIP),
    "c10_l2_k5": dict(dataset="cifar10", n=1000, d=32, seed=14,
                      argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5", "--dist_type",
                            "euclidean"] + SKIP),
    "c10_l2_k1_discrete_nd1": dict(dataset="cifar10", n=1000, d=32, seed=15,
                                   argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "1",
                                         "--dist_type", "euclidean", "--use_discrete_for_text", "--normalize_d1",
                                         "--noise_level", "0.2"] + SKIP),
    "c10_cos_k5_subset": dict(dataset="cifar10", n=1000, d=32, seed=16,
                              argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5",
                                    "--compr_dataset_size_limit", "300", "--seed", "3"] + SKIP),
    "c10_l2_k50_subset_discrete": dict(dataset="cifar10", n=1000, d=32, seed=17,
                                       argv=["--dataset", "cifar10", "--noise_type", "symmetric", "--knn_k", "50",
                                             "--dist_type", "euclidean", "--compr_dataset_size_limit", "300",
                                             "--use_discrete_for_text", "--batch_size", "96"] + SKIP),
    "c10_cos_k5_prompt_only_beta": dict(dataset="cifar10", n=600, d=32, seed=18, prefix="An image of the ",
                                        argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5",
                                              "--custom_cifar_prompt", "An image of the ", "--ablation", "only_beta",
                                              "--data_seed", "2"]),
    "c10_cos_k5_mmbaseline_skiptrain": dict(dataset="cifar10", n=600, d=32, seed=19,
                                            argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5",
                                                  "--ablation", "multimodal_baseline", "--skip_train",
                                                  "--subset_val_set", "40"]),
    "c10_cos_k5_openclip_branch": dict(dataset="cifar10", n=600, d=32, seed=20,
                                       argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5",
                                             "--clip_model", "cc3m_clip_from_scratch"] + SKIP),
    # the remaining ablation branches of run_lemon.py:341-384, full protocol (hyper-parameter search with force_zero /
    # force_one, d_1 zeroed for 'd1')
    # (planted seeds chosen so that the REFERENCE survives its own LBFGS polish, lib/metrics/utils.py:157-165: from the
    # start point [10]*6 it overflows exp(-tau*D) on many planted sets and then dies in fminbound -- DESIGN.md section 7;
    # LEMON_GOLDEN_ABL_SEED overrides the seed while searching for one)
    **{f"c10_cos_k5_abl_{a}": dict(dataset="cifar10", n=500, d=32, seed=int(os.environ.get("LEMON_GOLDEN_ABL_SEED", sd)),
                                   argv=["--dataset", "cifar10", "--noise_type", "asymmetric", "--knn_k", "5", "--ablation", a])
       for a, sd in (("d1", 61), ("tau_1_2", 41), ("beta", 70), ("tau_1", 43), ("tau_2", 93), ("gamma", 83))},
    # cifar10_full: train / val from an 80 / 20 split of the training set, test = the dataset's own test split with its own
    # noise vector (lib/datasets/utils.py:374-391)
    "c10full_cos_k5": dict(dataset="cifar10_full", n=800, n_test=150, d=32, seed=51,
                           argv=["--dataset", "cifar10_full", "--noise_type", "symmetric", "--knn_k", "5", "--data_seed", "2"] + SKIP),
    # 'real noise' CSV dataset through the real get_large_scale_dataset / LargeScaleDataset + image files on disk
    "cars_cos_k5_real": dict(dataset="stanford_cars", n=400, d=32, seed=50,
                             argv=["--dataset", "stanford_cars", "--noise_type", "real", "--noise_level", "0", "--real_dataset",
                                   "--knn_k", "5", "--data_seed", "1"]),
    # caption dataset (mostly unique text rows; DB = random subset smaller than train => mixed in_db)
    "coco_cos_k5_cat_subset": dict(dataset="mscoco", n=1000, d=32, seed=21,
                                   argv=["--dataset", "mscoco", "--noise_type", "cat", "--knn_k", "5",
                                         "--compr_dataset_size_limit", "400"] + SKIP),
    "coco_l2_k5_random_discrete": dict(dataset="mscoco", n=1000, d=32, seed=22,
                                       argv=["--dataset", "mscoco", "--noise_type", "random", "--knn_k", "5",
                                             "--dist_type", "euclidean", "--use_discrete_for_text",
                                             "--compr_dataset_size_limit", "400", "--seed", "1"] + SKIP),
    "coco_cos_k5_noun_full": dict(dataset="mscoco", n=800, d=32, seed=23,
                                  argv=["--dataset", "mscoco", "--noise_type", "noun", "--knn_k", "5", "--noise_level",
                                        "0.3"]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--disc", action="store_true", help="(re)generate only the discrepancy-baseline fixtures")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    state = {}
    dsu = install_stubs(state)
    work = tempfile.mkdtemp(prefix="lemon_loop_")
    try:
        for name, cfg in DISC_CASES.items():
            if a.only and a.only != name:
                continue
            fx = run_disc_case(name, cfg, dsu, state, work)
            print(f"disc_{name:31s} ssets={list(fx['ssets'])} n_db={len(fx['db_img'])} auroc={str(fx['auroc'])}", flush=True)
        for name, cfg in CASES.items():
            if a.disc or (a.only and a.only != name):
                continue
            fx = run_case(name, cfg, dsu, state, work)
            nbytes = os.path.getsize(os.path.join(OUT, f"loop_{name}.npz"))
            print(f"{name:36s} ssets={list(fx['ssets'])} n_db={len(fx['db_img'])} files={list(fx['out_files'])} {nbytes} B",
                  flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
