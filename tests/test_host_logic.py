"""CPU suite: the product's host logic against reference-generated golden vectors, and the
C-ABI library's export table (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pandas as pd
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(G.rstrip("/").rsplit("/", 1)[0])


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_label_sets_and_constants():
    from lemon_amd import datasets as ds
    meta = json.load(open(os.path.join(G, "dataset_meta.json")))
    assert ds.cifar10_labels.tolist() == meta["labels"]["cifar10"] and len(ds.cifar10_labels) == 10
    assert ds.cifar100_labels.tolist() == meta["labels"]["cifar100"] and len(ds.cifar100_labels) == 100
    assert len(ds.mini_imagenet_labels) == meta["class_num_dict"]["mini_imagenet"] == 100
    assert len(ds.stanford_cars_labels) == meta["class_num_dict"]["stanford_cars"] == 196
    assert ds.CLIP_MEAN == meta["CLIP_MEAN"] and ds.CLIP_STD == meta["CLIP_STD"]


@pytest.mark.parametrize("dataset", ["cifar10", "cifar100"])
def test_label_noise_streams_match_reference(dataset):
    from lemon_amd import datasets as ds
    g = load("noise_labels.npz")
    y = g[f"{dataset}_y"]
    for seed in (0, 1, 2):
        for lvl in (0.2, 0.4):
            for kind in ("asymmetric", "symmetric"):
                got = ds.add_noisy_labels(dataset, kind, lvl, seed, list(y))
                assert np.array_equal(got, g[f"{dataset}_{kind}_{seed}_{lvl}"]), (kind, seed, lvl)
    got = ds.add_noisy_labels(dataset, "asymmetric", 0.4, 0, list(y))
    assert 0.35 < (got != y).mean() < 0.45
    assert set(np.unique((got - y) % ds.class_num_dict[dataset])) == {0, 1}     # pair flip: c -> c+1


def test_cat_noise_on_cifar_raises_like_the_reference():
    from lemon_amd import datasets as ds
    assert int(load("noise_labels.npz")["cat_raises"]) == 1
    with pytest.raises(NotImplementedError):
        ds.add_noisy_labels("cifar100", "cat", 0.4, 0, [0, 1, 2])


def test_split_indices_match_reference():
    import hashlib
    from lemon_amd import datasets as ds
    g = load("splits.npz")
    for seed in (0, 1, 2):
        tr, va, te = ds.split_80_10_10(50000, seed)
        assert (len(tr), len(va), len(te)) == (40000, 5000, 5000)
        shas = [hashlib.sha256(np.ascontiguousarray(a.astype(np.int64)).tobytes()).hexdigest() for a in (tr, va, te)]
        assert shas == g[f"sha_{seed}"].tolist()
        assert np.array_equal(np.stack([tr[:16], va[:16], te[:16]]), g[f"head_{seed}"])
    tr, va, te = ds.split_80_10_10(50000, 0)
    assert np.array_equal(tr, g["train_0"]) and np.array_equal(va, g["val_0"]) and np.array_equal(te, g["test_0"])


def test_caption_noise_matches_reference():
    from lemon_amd import datasets as ds
    g = load("noise_captioning.npz")
    for seed in (0, 7):
        d = ds.random_noise_dict(50, 0.4, seed)
        assert np.array_equal(np.array(list(d.keys())), g[f"random_{seed}_keys"])
        assert np.array_equal(np.array(list(d.values())), g[f"random_{seed}_vals"])
        assert all(k != v for k, v in d.items())
    flat, lens = g["cats_flat"], g["cats_len"]
    cats, p = [], 0
    for n in lens:
        cats.append(flat[p:p + n].tolist()); p += n
    for seed in (0, 3):
        d = ds.calc_noise_by_integer_matching(np.array(cats, dtype=object), 0.4, seed)
        assert np.array_equal(np.array(list(d.keys())), g[f"match_{seed}_keys"])
        assert np.array_equal(np.array(list(d.values())), g[f"match_{seed}_vals"])
    frame = pd.DataFrame({"sentence": [f"caption {i % 37}" for i in range(60)]}, index=np.arange(100, 160))
    noised = ds.noise_given_dict(frame, ds.random_noise_dict(60, 0.3, 1))
    assert np.array_equal(np.array([int(s.split()[1]) for s in noised["sentence"]]), g["given_sentence_id"])
    assert np.array_equal(noised["is_mislabel"].values.astype(np.uint8), g["given_is_mislabel"])


# ------------------------------------------------------------------ the C-ABI boundary
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lemon_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lemon_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from lemon_amd import _lib
    names = _declared_symbols()
    assert sorted(_lib.EXPORTS) == names, "the ctypes binding and include/lemon_hip.h disagree"
    assert os.path.exists(_lib.SO_PATH), "liblemon_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/lemon_hip.h but not exported"
    # signatures are plain C: no torch / C++ types leak into the ABI
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lemon_hip.h")).read(), flags=re.S)
    assert "torch" not in hdr and "at::" not in hdr and "std::" not in hdr and "hipStream_t" not in hdr


def test_library_contains_gfx950_code_object():
    from lemon_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_scan_f32" in blob


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under lemon_amd/ or include/ may reference it."""
    bad = []
    for base in ("lemon_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"\boracle\b", txt) and "oracle" in txt.replace("CPU oracle", "").replace("the oracle", "").replace("oracle's", ""):
                        if re.search(r"(import|from|include|CDLL|dlopen).*oracle", txt):
                            bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_host_inputs_are_refused_without_gpu_fallback():
    import torch
    from lemon_amd import ops, _lib
    with pytest.raises((_lib.LemonHipError, TypeError)):
        ops.normalize_vectors(torch.zeros(3, 4))        # CPU tensor: no silent CPU path
    with pytest.raises(TypeError):
        ops.normalize_vectors(np.zeros((3, 4), np.float32))


def test_gemm_tuning_results_file_is_well_formed():
    # lemon_amd/tuning.py: recorded hipBLASLt solutions for the headline encoder shapes
    from lemon_amd import tuning
    lines = [l.strip().split(",") for l in open(tuning.RESULTS) if l.strip()]
    validators = {l[1]: l[2] for l in lines if l[0] == "Validator"}
    assert validators.get("GCN_ARCH_NAME", "").startswith("gfx950")
    ops = [l for l in lines if l[0] != "Validator"]
    assert ops and all(len(l) == 4 and l[0].startswith("Gemm") and float(l[3]) > 0 for l in ops)
    keys = {l[1] for l in ops}
    assert "tn_2304_50000_768_ld_768_768_2304" in keys      # ViT-B/32 QKV projection at encoder batch 1000


@pytest.mark.parametrize("h,w,oh,ow", [(32, 32, 224, 224), (48, 64, 224, 298), (300, 200, 336, 224), (500, 375, 298, 224),
                                       (37, 91, 224, 550), (640, 480, 298, 224)])
def test_pil_bicubic_tables_reproduce_pil_resize(h, w, oh, ow):
    # lemon_amd/data.py::pil_bicubic_tables (what lemon_preprocess_u8 consumes) against PIL itself
    from PIL import Image
    from lemon_amd.data import PIL_PRECISION_BITS, pil_bicubic_tables

    def axis(img, out_size):          # resample axis 1 of a uint8 [H, W, C] array
        kk, b = pil_bicubic_tables(img.shape[1], out_size)
        out = np.zeros((img.shape[0], out_size, img.shape[2]), np.uint8)
        for xx in range(out_size):
            x0, n = b[xx]
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (img[:, x0:x0 + n].astype(np.int64) * kk[xx, :n][None, :, None]).sum(1)
            out[:, xx] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        return out

    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = axis(axis(img, ow).transpose(1, 0, 2), oh).transpose(1, 0, 2)      # horizontal pass, then vertical
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(got, ref)
