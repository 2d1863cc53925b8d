"""Dataset readers and preprocessing for the run_lemon surface, without torchvision.

Mirrors (behaviour, not code) of:
  generic_transform                      lib/datasets/utils.py:163-170  (Resize(224, bicubic) -> CenterCrop(224)
                                         -> ToTensor -> Normalize(CLIP_MEAN, CLIP_STD))
  get_dataset('cifar10'|'cifar100'|...)  lib/datasets/utils.py:350-430  (torchvision CIFAR pickles, 80/10/10 split)
  NoisyCombinedDataset                   lib/datasets/dataloader.py:16-30   -> (x, clean, noisy)
  get_captioning_dataset / CaptioningDataset   lib/datasets/utils.py:275-323, dataloader.py:167-198
  get_large_scale_dataset / LargeScaleDataset  lib/datasets/utils.py:325-347, dataloader.py:113-133
Datasets are read from LOCAL paths only (no download: there is no network).  A synthetic class dataset
(`dataset_root='synthetic:N'`) stands in when no data is present.
"""
import os
import pickle
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import datasets as ds


# ------------------------------------------------------------------------------ preprocessing
def generic_transform(img, size=224):
    """PIL RGB image -> float32 [3,size,size], CLIP-normalised.  Same steps as torchvision's
    Resize(shorter side, BICUBIC on the PIL image) / CenterCrop / ToTensor / Normalize."""
    from PIL import Image
    w, h = img.size
    if (w <= h and w != size) or (h <= w and h != size):
        if w <= h:
            nw, nh = size, int(size * h / w)
        else:
            nw, nh = int(size * w / h), size
        img = img.resize((nw, nh), Image.BICUBIC)
        w, h = nw, nh
    left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(ds.CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(ds.CLIP_STD).view(3, 1, 1)
    return (x - mean) / std


class ImageLabelSet:
    """(x, clean, noisy) triples like NoisyCombinedDataset; `images` is uint8 [N,H,W,3] or a list of
    file paths; labels are ints (class datasets) or strings (captions)."""

    def __init__(self, images, clean, noisy, image_size=224, workers=8):
        assert len(images) == len(clean) == len(noisy)
        self.images, self.clean, self.noisy = images, clean, noisy
        self.image_size, self.workers = image_size, workers

    def __len__(self):
        return len(self.noisy)

    def subset(self, idx):
        pick = (lambda a: a[idx]) if isinstance(self.images, np.ndarray) else (lambda a: [a[i] for i in idx])
        lab = lambda a: a[idx] if isinstance(a, np.ndarray) else [a[i] for i in idx]
        return ImageLabelSet(pick(self.images), lab(self.clean), lab(self.noisy), self.image_size, self.workers)

    def _load(self, i):
        from PIL import Image
        item = self.images[i]
        img = Image.fromarray(item) if isinstance(item, np.ndarray) else Image.open(item).convert("RGB")
        return generic_transform(img, self.image_size)

    def batches(self, batch_size, lo=0, hi=None):
        """Yield (pixel_values [B,3,S,S] f32, clean[B], noisy[B]) in order (never shuffled, last batch
        short: SURVEY Appendix B.6).  PIL work runs in a thread pool (the reference forks 8 DataLoader
        workers, run_lemon.py:129-131; threads avoid fork-after-HIP-init)."""
        hi = len(self) if hi is None else hi
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            for s in range(lo, hi, batch_size):
                idx = range(s, min(hi, s + batch_size))
                px = torch.stack(list(pool.map(self._load, idx)))
                sl = slice(s, min(hi, s + batch_size))
                yield px, self.clean[sl], self.noisy[sl]


class SyntheticPixelSet(ImageLabelSet):
    """Seeded random 'images' generated per batch (no PIL): for runs without any dataset on disk."""

    def __init__(self, n, clean, noisy, image_size=224, seed=0):
        self.n, self.clean, self.noisy, self.image_size, self.seed = n, clean, noisy, image_size, seed
        self.images = None

    def __len__(self):
        return self.n

    def subset(self, idx):
        out = SyntheticPixelSet(len(idx), self.clean[idx], self.noisy[idx], self.image_size, self.seed)
        out.rows = (self.rows[idx] if hasattr(self, "rows") else np.asarray(idx))
        return out

    def batches(self, batch_size, lo=0, hi=None):
        hi = self.n if hi is None else hi
        rows = self.rows if hasattr(self, "rows") else np.arange(self.n)
        for s in range(lo, hi, batch_size):
            e = min(hi, s + batch_size)
            px = torch.stack([torch.randn(3, self.image_size, self.image_size,
                                          generator=torch.Generator().manual_seed(self.seed * 1_000_003 + int(r)))
                              for r in rows[s:e]])
            yield px, self.clean[s:e], self.noisy[s:e]


# ------------------------------------------------------------------------------ dataset factory
def _read_cifar(root, name):
    if name.startswith("cifar100"):
        with open(os.path.join(root, "cifar-100-python", "train"), "rb") as f:
            d = pickle.load(f, encoding="bytes")
        x, y = d[b"data"], np.array(d[b"fine_labels"])
    else:
        xs, ys = [], []
        for i in range(1, 6):
            with open(os.path.join(root, "cifar-10-batches-py", f"data_batch_{i}"), "rb") as f:
                d = pickle.load(f, encoding="bytes")
            xs.append(d[b"data"]); ys += list(d[b"labels"])
        x, y = np.concatenate(xs), np.array(ys)
    return np.ascontiguousarray(x.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)), y


def get_dataset(name, data_seed, percent_flips=0.40, flip_type="real", data_root="./data", image_size=224):
    """train/val/test ImageLabelSets for the datasets run_lemon.py accepts (run_lemon.py:37-38,105-106).
    `data_root='synthetic:N'` builds an N-sample synthetic class dataset with the named dataset's labels."""
    if name in ("cifar10", "cifar100"):
        C = ds.class_num_dict[name]
        if str(data_root).startswith("synthetic"):
            n = int(str(data_root).split(":")[1]) if ":" in str(data_root) else 5000
            y = np.random.RandomState(data_seed).randint(0, C, n)
            images = None
        else:
            images, y = _read_cifar(data_root, name)
            n = len(y)
        noisy = np.asarray(ds.add_noisy_labels(name, flip_type, percent_flips, data_seed, list(y), data_root))
        tr, va, te = ds.split_80_10_10(n, data_seed)
        if images is None:
            full = SyntheticPixelSet(n, y, noisy, image_size, data_seed)
        else:
            full = ImageLabelSet(images, y, noisy, image_size)
        return full.subset(tr), full.subset(va), full.subset(te)
    if name in ("mscoco", "flickr30k", "mimiccxr_caption", "mmimdb", "cc3m"):
        import pandas as pd
        df = pd.read_pickle(os.path.join(data_root, "multimodal_mislabel_split.pkl"))
        if "restval" in df.split:      # quirk kept: tests the Series INDEX
            df.loc[df.split == "restval", "split"] = "train"
        if "path" not in df:
            df["path"] = [os.path.join(data_root, *(p for p in (r.get("filepath", ""), r["filename"]) if p))
                          for _, r in df.iterrows()]
        out = []
        for split in ("train", "val", "test"):
            part = df.query(f'split == "{split}"')
            if flip_type == "random":
                nd = ds.random_noise_dict(len(part), percent_flips, data_seed)
            elif flip_type == "noun":
                nd = ds.calc_noise_by_integer_matching(part["nouns_int"].values, percent_flips, data_seed)
            elif flip_type == "cat":
                nd = ds.calc_noise_by_integer_matching(part["cat_labels"].values, percent_flips, data_seed)
            else:
                raise NotImplementedError(flip_type)
            part = ds.noise_given_dict(part, nd)
            out.append(ImageLabelSet(list(part["path"]), list(part["gold_sentence"]), list(part["sentence"]), image_size))
        return tuple(out)
    if name in ("stanford_cars", "mini_imagenet"):
        import pandas as pd
        from sklearn.model_selection import train_test_split
        assert flip_type == "real"
        df = pd.read_csv(os.path.join(data_root, "multimodal_mislabel_split.csv"))
        if "path" not in df:
            df["path"] = [os.path.join(data_root, f) for f in df["filename"]]
        trv, te = train_test_split(df.index, random_state=data_seed, train_size=0.75, stratify=df.is_clean)
        tr, va = train_test_split(trv, random_state=data_seed, train_size=0.5 / 0.75, stratify=df.loc[trv].is_clean)
        out = []
        for idx in (tr, va, te):
            part = df.loc[sorted(idx)]
            noisy = part["label"].values
            clean = np.where(part["is_clean"].values, noisy, noisy - 1)   # dataloader.py:130-131
            out.append(ImageLabelSet(list(part["path"]), clean, noisy, image_size))
        return tuple(out)
    raise NotImplementedError(name)
