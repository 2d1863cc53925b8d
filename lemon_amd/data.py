"""Dataset readers and preprocessing for the run_lemon surface, without torchvision.

Mirrors (behaviour, not code) of:
  generic_transform                      lib/datasets/utils.py:163-170  (Resize(224, bicubic) -> CenterCrop(224)
                                         -> ToTensor -> Normalize(CLIP_MEAN, CLIP_STD))
  get_dataset('cifar10'|'cifar100'|...)  lib/datasets/utils.py:350-430  (torchvision CIFAR pickles, 80/10/10 split)
  NoisyCombinedDataset                   lib/datasets/dataloader.py:16-30   -> (x, clean, noisy)
  get_captioning_dataset / CaptioningDataset   lib/datasets/utils.py:275-323, dataloader.py:167-198
  get_large_scale_dataset / LargeScaleDataset  lib/datasets/utils.py:325-347, dataloader.py:113-133
Datasets are read from LOCAL paths only (no download: there is no network).  A synthetic class dataset
(`dataset_root='synthetic:N'`: seeded random uint8 images with the dataset's label set) stands in when no data is present.
"""
import functools
import math
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


# ------------------------------------------------------------------------------ preprocessing on the GPU
PIL_PRECISION_BITS = 22      # Pillow Resample.c: 32 - 8 - 2


def _bicubic(x):
    a = -0.5                 # Pillow's bicubic_filter
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def pil_bicubic_tables(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for BICUBIC (Resample.c), restated: for every
    output index the first input index, the tap count and the 22-bit fixed-point taps.  in == out -> the
    identity table (Pillow skips the pass; tap 1<<22 reproduces the pixel exactly)."""
    if in_size == out_size:
        kk = np.full((out_size, 1), 1 << PIL_PRECISION_BITS, np.int32)
        bounds = np.stack([np.arange(out_size), np.ones(out_size, np.int64)], 1).astype(np.int32)
        return kk, bounds
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


@functools.lru_cache(maxsize=16)
def _gpu_transform_plan(h, w, size, device_index):
    """Resize geometry of generic_transform for an h x w image + the cropped tap tables on the device."""
    if (w <= h and w != size) or (h <= w and h != size):
        nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    else:
        nw, nh = w, h
    if nw < size or nh < size:
        raise ValueError(f"image {h}x{w} resizes to {nh}x{nw}, smaller than the {size}x{size} crop")
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    kk_h, b_h = pil_bicubic_tables(w, nw)
    kk_v, b_v = pil_bicubic_tables(h, nh)
    kk_h, b_h, kk_v, b_v = kk_h[left:left + size], b_h[left:left + size], kk_v[top:top + size], b_v[top:top + size]
    rows_per_block = 16
    while True:     # input rows one block's vertical windows span; shrink the block until the tile fits LDS
        spans = [int(b_v[min(y0 + rows_per_block, size) - 1].sum() - b_v[y0, 0]) for y0 in range(0, size, rows_per_block)]
        # (the kernel's LDS: the uint8 tile <= 56 KB, and the block's vertical windows, rows x (2 + taps) <= 512 ints)
        fits = max(spans) * size * 3 <= 56 * 1024 and rows_per_block * (2 + kk_v.shape[1]) <= 512
        if fits or rows_per_block == 1:
            break
        rows_per_block //= 2
    if not fits:
        raise ValueError(f"image {h}x{w}: vertical window of {max(spans)} rows / {kk_v.shape[1]} taps does not fit the LDS tile")
    dev = torch.device("cuda", device_index)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(kk_h=t(kk_h), b_h=t(b_h), kk_v=t(kk_v), b_v=t(b_v), ks_h=kk_h.shape[1], ks_v=kk_v.shape[1],
                max_rows=max(spans), rows_per_block=rows_per_block)


class PatchOperand:
    """The patch rows of a preprocessed image batch as the tile-major fp16 split operand of lemon_linear_f16x3t
    (lemon_preprocess_u8_f16x3t): `at` flat fp16, rows = batch * n_patches (padded to 128), k = 3 * patch^2."""

    def __init__(self, at, batch, n_patches, k):
        self.at, self.batch, self.n_patches, self.k = at, batch, n_patches, k
        self.is_cuda, self.device = True, at.device


def patch_operand_supported(patch, size=224):
    return patch > 0 and patch % 4 == 0 and size % 4 == 0 and (3 * patch * patch) % 32 == 0


def gpu_transform_batch(images_u8, size=224, patch=0, operand=False):
    """generic_transform for a uint8 CUDA batch [B,H,W,3] in one HIP kernel (lemon_preprocess_u8):
    -> float32 [B,3,size,size], bit-identical to the PIL + torch pipeline; with patch=P the same values
    in patch-major order [B, (size/P)^2, 3*P*P] (the ViT patch embedding then is one GEMM); with operand=True (and
    patch_operand_supported(P)) a PatchOperand: the same rows already split for the hand-written GEMM."""
    import ctypes
    from . import _lib
    from .ops import ptr, stream_ptr
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3] == 3
    x = images_u8.contiguous()
    B, H, W, _ = x.shape
    plan = _gpu_transform_plan(H, W, size, x.device.index or 0)
    mean = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in ds.CLIP_MEAN])
    std = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in ds.CLIP_STD])
    lib = _lib.load()
    if operand:
        assert patch_operand_supported(patch, size)
        nP, K = (size // patch) ** 2, 3 * patch * patch
        rows = (B * nP + 127) // 128 * 128
        at = torch.empty((rows * K * 2,), dtype=torch.float16, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.lemon_preprocess_u8_f16x3t(ptr(x), B, H, W, ptr(plan["kk_h"]), ptr(plan["b_h"]), plan["ks_h"],
                                                      ptr(plan["kk_v"]), ptr(plan["b_v"]), plan["ks_v"], size, plan["max_rows"],
                                                      plan["rows_per_block"], mean, std, int(patch), ptr(at), stream_ptr(x.device)),
                       "lemon_preprocess_u8_f16x3t")
        return PatchOperand(at, B, nP, K)
    out = torch.empty((B, 3, size, size) if not patch else (B, (size // patch) ** 2, 3 * patch * patch),
                      dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.lemon_preprocess_u8(ptr(x), B, H, W, ptr(plan["kk_h"]), ptr(plan["b_h"]), plan["ks_h"],
                                           ptr(plan["kk_v"]), ptr(plan["b_v"]), plan["ks_v"], size, plan["max_rows"],
                                           plan["rows_per_block"], mean, std, int(patch), ptr(out), stream_ptr(x.device)),
                   "lemon_preprocess_u8")
    return out


class ImageLabelSet:
    """(x, clean, noisy) triples like NoisyCombinedDataset / CaptioningDataset.  `images` is one of
      * uint8 [N,H,W,3] in memory (CIFAR pickles, `pixels.npy`): generic_transform runs on the GPU per batch;
      * float32 [N,...] in memory: pixel tensors that are ALREADY what the model consumes (a preprocessed cache);
        passed through unchanged;
      * a list of file paths: PIL decode + generic_transform in a thread pool.
    Labels are ints (class datasets) or strings (captions)."""

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

    def batches(self, batch_size, lo=0, hi=None, device=None):
        """Yield (pixel_values [B,3,S,S] f32, clean[B], noisy[B]) in order (never shuffled, last batch
        short: SURVEY Appendix B.6).  PIL work runs in a thread pool (the reference forks 8 DataLoader
        workers, run_lemon.py:129-131; threads avoid fork-after-HIP-init)."""
        hi = len(self) if hi is None else hi
        if isinstance(self.images, np.ndarray) and self.images.dtype == np.float32:
            for s in range(lo, hi, batch_size):
                sl = slice(s, min(hi, s + batch_size))
                yield torch.from_numpy(np.ascontiguousarray(self.images[sl])), self.clean[sl], self.noisy[sl]
            return
        if device is not None and torch.device(device).type == "cuda" and isinstance(self.images, np.ndarray) \
                and self.images.dtype == np.uint8 and self.images.ndim == 4:
            # in-memory uint8 arrays (CIFAR): 3 KB per image cross PCIe instead of 602 KB, and the
            # bicubic up-sampling runs in lemon_preprocess_u8 (bit-identical to the PIL path below)
            for s in range(lo, hi, batch_size):
                sl = slice(s, min(hi, s + batch_size))
                u8 = torch.from_numpy(np.ascontiguousarray(self.images[sl])).to(device, non_blocking=True)
                yield gpu_transform_batch(u8, self.image_size), self.clean[sl], self.noisy[sl]
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            for s in range(lo, hi, batch_size):
                idx = range(s, min(hi, s + batch_size))
                px = torch.stack(list(pool.map(self._load, idx)))
                sl = slice(s, min(hi, s + batch_size))
                yield px, self.clean[sl], self.noisy[sl]


# ------------------------------------------------------------------------------ dataset factory
def synthetic_caption_frame(n, seed, n_cat=80, image_hw=32):
    """Stand-in for a caption dataset's `multimodal_mislabel_split.pkl` (lib/datasets/utils.py:275-323) when no data is
    present: n rows with the columns the reader uses -- split (train/val/test/restval in Karpathy-like proportions;
    restval rows are dropped by the reference's no-op remap), a UNIQUE sentence per row (a few exact
    duplicates, as in COCO), `cat_labels` / `nouns_int` id lists (some rows without categories) -- plus in-memory
    uint8 images [n, hw, hw, 3] whose pattern depends on the first category, so an encoder sees structure."""
    import pandas as pd
    rs = np.random.RandomState(seed)
    cats = [sorted(set(rs.randint(0, n_cat, rs.randint(1, 4)).tolist())) for _ in range(n)]
    for j in range(0, n, 53):
        cats[j] = []
    nouns = [sorted(set(rs.randint(0, 400, rs.randint(1, 6)).tolist())) for _ in range(n)]
    first = np.array([c[0] if c else rs.randint(0, n_cat) for c in cats])
    words = ["red", "small", "two", "wooden", "old", "bright", "open", "tall", "wet", "quiet", "busy", "empty"]
    sent = [f"a {words[i % 12]} {words[(i // 12) % 12]} scene number {i} showing object {first[i]} near thing {nouns[i][0]}"
            for i in range(n)]
    for j in range(7, n, 97):
        sent[j] = sent[j - 7]
    u = rs.rand(n)
    split = np.where(u < 0.66, "train", np.where(u < 0.70, "val", np.where(u < 0.74, "test", "restval"))).astype(object)
    cocoid = 100000 + rs.permutation(3 * n)[:n]
    pat = rs.randint(0, 256, (n_cat, image_hw, image_hw, 3)).astype(np.int16)
    px = np.clip(pat[first] + rs.randint(-64, 65, (n, image_hw, image_hw, 3)), 0, 255).astype(np.uint8)
    df = pd.DataFrame({"split": split, "filepath": "synthetic", "filename": [f"{c}.jpg" for c in cocoid], "sentence": sent,
                       "cat_labels": cats, "nouns_int": nouns}, index=cocoid)
    return df, px


def _read_cifar(root, name, train=True):
    """The CIFAR python pickles torchvision's CIFAR10 / CIFAR100(train=...) classes unpickle (lib/datasets/utils.py:356-386):
    uint8 [N, 32, 32, 3] images and the (fine) labels."""
    if name.startswith("cifar100"):
        with open(os.path.join(root, "cifar-100-python", "train" if train else "test"), "rb") as f:
            d = pickle.load(f, encoding="bytes")
        x, y = d[b"data"], np.array(d[b"fine_labels"])
    else:
        xs, ys = [], []
        for fn in ([f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]):
            with open(os.path.join(root, "cifar-10-batches-py", fn), "rb") as f:
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
            rs = np.random.RandomState(data_seed)
            y = rs.randint(0, C, n)
            # seeded random CIFAR-shaped uint8 images: the same code path (and GPU preprocessing) as the real pickles
            images = rs.randint(0, 256, (n, 32, 32, 3), dtype=np.uint8)
        else:
            images, y = _read_cifar(data_root, name)
            n = len(y)
        noisy = np.asarray(ds.add_noisy_labels(name, flip_type, percent_flips, data_seed, list(y), data_root))
        tr, va, te = ds.split_80_10_10(n, data_seed)
        full = ImageLabelSet(images, y, noisy, image_size)
        return full.subset(tr), full.subset(va), full.subset(te)
    if name in ("cifar10_full", "cifar100_full"):
        # lib/datasets/utils.py:374-391: train / val = an 80 / 20 split of the training set, test = the dataset's own test
        # split, each with its own noise vector drawn with the same seed
        C = ds.class_num_dict[name]
        if str(data_root).startswith("synthetic"):
            n = int(str(data_root).split(":")[1]) if ":" in str(data_root) else 5000
            rs = np.random.RandomState(data_seed)
            y, y_te = rs.randint(0, C, n), rs.randint(0, C, max(n // 5, 1))
            images = rs.randint(0, 256, (n, 32, 32, 3), dtype=np.uint8)
            images_te = rs.randint(0, 256, (len(y_te), 32, 32, 3), dtype=np.uint8)
        else:
            images, y = _read_cifar(data_root, name, True)
            images_te, y_te = _read_cifar(data_root, name, False)
        noisy = np.asarray(ds.add_noisy_labels(name, flip_type, percent_flips, data_seed, list(y), data_root))
        noisy_te = np.asarray(ds.add_noisy_labels(name, flip_type, percent_flips, data_seed, list(y_te), data_root))
        tr, va = ds.split_80_20(len(y), data_seed)
        full = ImageLabelSet(images, y, noisy, image_size)
        return full.subset(tr), full.subset(va), ImageLabelSet(images_te, y_te, noisy_te, image_size)
    if name in ("mscoco", "flickr30k", "mimiccxr_caption", "mmimdb", "cc3m"):
        import pandas as pd
        pixels = None
        if str(data_root).startswith("synthetic"):
            n = int(str(data_root).split(":")[1]) if ":" in str(data_root) else 5000
            df, pixels = synthetic_caption_frame(n, data_seed)
        else:
            df = pd.read_pickle(os.path.join(data_root, "multimodal_mislabel_split.pkl"))
            # optional pre-decoded images aligned with the frame's rows: uint8 [N,H,W,3] (GPU preprocessing) or
            # float32 [N,...] (already preprocessed); replaces per-file JPEG decoding
            if os.path.exists(os.path.join(data_root, "pixels.npy")):
                pixels = np.load(os.path.join(data_root, "pixels.npy"), mmap_mode="r")
                assert len(pixels) == len(df), "pixels.npy must have one entry per frame row"
        if pixels is not None:         # position in pixels.npy = position in the frame AS LOADED (before any row is dropped)
            df = df.assign(_row=np.arange(len(df)))
        if "restval" in df.split:      # quirk kept: tests the Series INDEX
            df.loc[df.split == "restval", "split"] = "train"
        if name == "mimiccxr_caption":
            df = df[df.sentence.str.len() > 0]       # lib/datasets/utils.py:293
        if pixels is None and "path" not in df:
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
            images = np.ascontiguousarray(pixels[part["_row"].values]) if pixels is not None else list(part["path"])
            out.append(ImageLabelSet(images, list(part["gold_sentence"]), list(part["sentence"]), image_size))
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
