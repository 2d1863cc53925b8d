"""CPU: the file-based dataset readers of lemon_amd/data.py on real files.

  * stanford_cars / mini_imagenet CSV reader + `real_label = noisy - 1` against the fixture of a REFERENCE run that went
    through the real get_large_scale_dataset / LargeScaleDataset (lib/datasets/utils.py:325-347, dataloader.py:113-133;
    tools/make_golden_loop.py case cars_cos_k5_real): same split membership and order, same labels, same decoded images;
  * the CIFAR python-pickle reader on files in the published CIFAR-10 / CIFAR-100 layout (what torchvision's CIFAR10 /
    CIFAR100 classes, lib/datasets/utils.py:356-372, unpickle): HWC order, label order over the five batches, the golden
    80/10/10 split and noise vectors;
  * caption frames with `pixels.npy`: rows dropped by the mimiccxr empty-sentence filter (lib/datasets/utils.py:293) must not
    shift the image <-> caption alignment (round-2 advisor finding)."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest
import torch

from tests.loopfx import LoopCase


def test_large_scale_csv_reader_matches_reference_run(monkeypatch, tmp_path):
    from lemon_amd import data
    from tests import planted
    c = LoopCase("cars_cos_k5_real")
    extra = planted.install(c, monkeypatch, tmp_path)
    root = extra[extra.index("--data_root") + 1]
    a = c.argv
    seed = int(a[a.index("--data_seed") + 1])
    sets = data.get_dataset("stanford_cars", seed, percent_flips=0.0, flip_type="real", data_root=root)
    assert [len(s) for s in sets] == [len(c.fx[f"{s}_d_1"]) for s in ("train", "val", "test")]
    raw = c.fx["img_all"]
    for part, s in zip(sets, ("train", "val", "test")):
        assert np.array_equal(np.asarray(part.noisy, np.int64), c.fx[f"{s}_noisy"]), s
        assert np.array_equal(np.asarray(part.clean, np.int64), c.fx[f"{s}_clean"]), s       # noisy - 1 where not clean
        px = torch.cat([b[0] for b in part.batches(64)]).numpy()                             # PIL decode of the files
        # the reference's query embeddings of the split are normalize(planted vector of the SAME rows)
        q = c.fx[f"{s}_q_img"]
        nrm = px / np.maximum(np.linalg.norm(px.astype(np.float64), axis=1, keepdims=True), 1e-12)
        assert np.abs(nrm - q).max() <= 2e-7, s
        assert all(os.path.isfile(f) for f in part.images)
    assert len(raw) == sum(len(s) for s in sets)


def test_cifar_full_split_matches_reference_run(monkeypatch, tmp_path):
    """cifar10_full (lib/datasets/utils.py:374-391): train / val = 80 / 20 of the training set, test = the dataset's own test
    split with its own noise vector -- labels and membership against the reference's own run (loop_c10full_cos_k5.npz)."""
    from lemon_amd import data
    from tests import planted
    c = LoopCase("c10full_cos_k5")
    extra = planted.install(c, monkeypatch, tmp_path)
    a = c.argv
    seed = int(a[a.index("--data_seed") + 1])
    sets = data.get_dataset("cifar10_full", seed, percent_flips=0.4, flip_type="symmetric", data_root=extra[1])
    assert [len(s) for s in sets] == [len(c.fx[f"{s}_d_1"]) for s in ("train", "val", "test")] == [640, 160, 150]
    for part, s in zip(sets, ("train", "val", "test")):
        assert np.array_equal(np.asarray(part.noisy, np.int64), c.fx[f"{s}_noisy"]), s
        assert np.array_equal(np.asarray(part.clean, np.int64), c.fx[f"{s}_clean"]), s
    assert np.array_equal(sets[2].images, c.fx["img_test"])           # the test split is the test set, in file order


def _write_cifar(root, name, x, y):
    if name == "cifar100":
        os.makedirs(os.path.join(root, "cifar-100-python"))
        with open(os.path.join(root, "cifar-100-python", "train"), "wb") as f:
            pickle.dump({b"data": x, b"fine_labels": [int(v) for v in y], b"coarse_labels": [int(v) // 5 for v in y],
                         b"filenames": [b"x.png"] * len(y), b"batch_label": b"training batch 1 of 1"}, f)
    else:
        os.makedirs(os.path.join(root, "cifar-10-batches-py"))
        per = len(y) // 5
        for i in range(5):
            with open(os.path.join(root, "cifar-10-batches-py", f"data_batch_{i + 1}"), "wb") as f:
                pickle.dump({b"data": x[i * per:(i + 1) * per], b"labels": [int(v) for v in y[i * per:(i + 1) * per]],
                             b"filenames": [b"x.png"] * per, b"batch_label": b"training batch"}, f)


@pytest.mark.parametrize("name,C", [("cifar10", 10), ("cifar100", 100)])
def test_cifar_pickle_reader(tmp_path, name, C):
    """CIFAR's published layout: `data` is uint8 [N, 3072] = R plane, G plane, B plane of a 32x32 image, row-major."""
    from lemon_amd import data
    from lemon_amd import datasets as ds
    rs = np.random.RandomState(4)
    n = 500
    hwc = rs.randint(0, 256, (n, 32, 32, 3), dtype=np.uint8)
    y = rs.randint(0, C, n)
    _write_cifar(str(tmp_path), name, np.ascontiguousarray(hwc.transpose(0, 3, 1, 2).reshape(n, 3072)), y)
    images, labels = data._read_cifar(str(tmp_path), name)
    assert images.dtype == np.uint8 and images.shape == (n, 32, 32, 3) and images.flags["C_CONTIGUOUS"]
    assert np.array_equal(images, hwc) and np.array_equal(labels, y)
    tr, va, te = data.get_dataset(name, data_seed=1, percent_flips=0.4, flip_type="asymmetric", data_root=str(tmp_path))
    i_tr, i_va, i_te = ds.split_80_10_10(n, 1)
    noisy = np.asarray(ds.add_noisy_labels(name, "asymmetric", 0.4, 1, list(y)))
    for part, idx in ((tr, i_tr), (va, i_va), (te, i_te)):
        assert np.array_equal(part.images, hwc[idx]) and np.array_equal(part.clean, y[idx])
        assert np.array_equal(part.noisy, noisy[idx])
    px = next(tr.batches(4))[0]                           # PIL path on CPU: CHW float, CLIP-normalised
    assert px.shape == (4, 3, 224, 224) and px.dtype == torch.float32
    # the *_full variants also read the dataset's test split (test_batch / test)
    n_te = 120
    hwc_te = rs.randint(0, 256, (n_te, 32, 32, 3), dtype=np.uint8)
    y_te = rs.randint(0, C, n_te)
    sub = "cifar-100-python" if name == "cifar100" else "cifar-10-batches-py"
    with open(os.path.join(str(tmp_path), sub, "test" if name == "cifar100" else "test_batch"), "wb") as f:
        pickle.dump({b"data": np.ascontiguousarray(hwc_te.transpose(0, 3, 1, 2).reshape(n_te, 3072)),
                     (b"fine_labels" if name == "cifar100" else b"labels"): [int(v) for v in y_te]}, f)
    im_te, lab_te = data._read_cifar(str(tmp_path), name, train=False)
    assert np.array_equal(im_te, hwc_te) and np.array_equal(lab_te, y_te)
    ftr, fva, fte = data.get_dataset(name + "_full", data_seed=1, percent_flips=0.4, flip_type="asymmetric", data_root=str(tmp_path))
    i_tr, i_va = ds.split_80_20(n, 1)
    assert np.array_equal(ftr.images, hwc[i_tr]) and np.array_equal(fva.clean, y[i_va]) and np.array_equal(fte.images, hwc_te)
    assert np.array_equal(fte.noisy, np.asarray(ds.add_noisy_labels(name + "_full", "asymmetric", 0.4, 1, list(y_te))))


def test_pixels_stay_aligned_when_the_mimic_filter_drops_rows(tmp_path):
    from lemon_amd import data
    n = 60
    rs = np.random.RandomState(0)
    sent = [f"finding number {i}" for i in range(n)]
    for j in (3, 17, 40):
        sent[j] = ""                                      # reports without FINDINGS / IMPRESSION (utils.py:293)
    df = pd.DataFrame({"split": ["train"] * 40 + ["val"] * 10 + ["test"] * 10, "filename": [f"{i}.jpg" for i in range(n)],
                       "sentence": sent, "cat_labels": [[int(i % 5)] for i in range(n)]}, index=np.arange(1000, 1000 + n))
    px = np.zeros((n, 2, 2, 3), np.uint8)
    px[:, 0, 0, 0] = np.arange(n)                         # the image knows its frame row
    df.to_pickle(tmp_path / "multimodal_mislabel_split.pkl")
    np.save(tmp_path / "pixels.npy", px)
    sets = data.get_dataset("mimiccxr_caption", 0, percent_flips=0.0, flip_type="cat", data_root=str(tmp_path))
    seen = 0
    for part in sets:
        for img, gold in zip(part.images, part.clean):
            assert gold == f"finding number {int(img[0, 0, 0])}"
            seen += 1
    assert seen == n - 3
