"""Product-side twins of the stand-ins tools/make_golden_loop.py gave the REFERENCE run: a planted "CLIP" whose image
embedding is the vector carried as the pixel tensor and whose text embedding is a table row, and its tokenizer.  With
them `lemon_amd.run_lemon.main` consumes exactly the inputs the reference's run_lemon.py consumed for a loop fixture."""
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch


class PlantedCLIP(torch.nn.Module):
    def __init__(self, txt_table):
        super().__init__()
        self.register_buffer("table", torch.from_numpy(np.ascontiguousarray(txt_table)))
        d = txt_table.shape[1]
        self.cfg = SimpleNamespace(image_size=1, patch_size=1, embed_dim=d, context_length=8)
        self.context_length = 8

    def encode_image(self, pixel_values=None):
        return pixel_values.float().clone()

    def encode_text(self, input_ids=None, attention_mask=None, seq_len=None):
        return self.table[input_ids[:, 0]].clone()


class PlantedTokenizer:
    def __init__(self, prompt_ids, hf):
        self.prompt_ids, self.hf = prompt_ids, hf

    def __call__(self, texts, padding=None, truncation=None):
        ids = [[self.prompt_ids[t], 0] for t in texts]
        if self.hf:
            return {"input_ids": ids, "attention_mask": [[1, 0] for _ in ids]}
        return torch.tensor(ids, dtype=torch.long)


def install(case, monkeypatch, tmp_path):
    """Point lemon_amd.run_lemon at the fixture's planted model / data.  Returns the extra CLI args."""
    import lemon_amd.clip as clip
    import lemon_amd.data as data
    from lemon_amd import datasets as ds
    fx = case.fx
    table = fx["txt_table"]
    if case.is_caption:
        prompt_ids = {str(c): i for i, c in enumerate(fx["captions"])}
    else:
        labels = ds.LABEL_SETS[case.dataset]          # (cifar10_full shares cifar10's label set)
        prompt_ids = {str(fx["prefix"]) + l: i for i, l in enumerate(labels)}

    def factory(name, text_base_name=None, img_base=None, return_tokenizer=False, arch=None, bpe_path=None):
        model = PlantedCLIP(table)
        tok = PlantedTokenizer(prompt_ids, hf=(name == "huggingface_clip"))
        return (model, tok) if return_tokenizer else model

    monkeypatch.setattr(clip, "algorithm_class_from_scratch", factory)
    root = str(tmp_path / "data")
    os.makedirs(root, exist_ok=True)
    if case.is_caption:
        def unflat(flat, lens):
            out, p = [], 0
            for n in lens:
                out.append([int(v) for v in flat[p:p + n]])
                p += n
            return out
        df = pd.DataFrame({"split": fx["frame_split"].astype(object), "filepath": "synthetic",
                           "filename": fx["frame_filename"].astype(object), "sentence": fx["frame_sentence"].astype(object),
                           "cat_labels": unflat(fx["frame_cat_flat"], fx["frame_cat_len"]),
                           "nouns_int": unflat(fx["frame_noun_flat"], fx["frame_noun_len"])}, index=fx["frame_index"])
        df.to_pickle(os.path.join(root, "multimodal_mislabel_split.pkl"))
        np.save(os.path.join(root, "pixels.npy"), fx["img_all"])
    elif case.dataset in ("stanford_cars", "mini_imagenet"):
        # the CSV + image files the reference's get_large_scale_dataset / LargeScaleDataset read (lib/datasets/utils.py:325-347,
        # dataloader.py:113-133): lossless PNGs whose pixel bytes carry the planted vector, decoded by the same stand-in
        # transform the reference run was given
        from PIL import Image
        d = int(fx["d"])
        for vec, fn in zip(fx["img_all"], fx["csv_filename"]):
            path = os.path.join(root, str(fn))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            raw = np.asarray(vec, np.float32).tobytes()
            w = -(-len(raw) // 12)
            buf = np.zeros(4 * w * 3, np.uint8)
            buf[:len(raw)] = np.frombuffer(raw, np.uint8)
            Image.fromarray(buf.reshape(4, w, 3), "RGB").save(path, format="PNG")
        pd.DataFrame({"filename": fx["csv_filename"].astype(object), "label": fx["csv_label"],
                      "is_clean": fx["csv_is_clean"].astype(bool)}).to_csv(os.path.join(root, "multimodal_mislabel_split.csv"), index=False)
        monkeypatch.setattr(data, "generic_transform", lambda img, size=224: torch.from_numpy(
            np.frombuffer(np.asarray(img, np.uint8).tobytes()[:4 * d], np.float32).copy()))
    else:
        monkeypatch.setattr(data, "_read_cifar", lambda r, n, train=True: (fx["img_all"], fx["y_all"]) if train else (fx["img_test"], fx["y_test"]))
    return ["--data_root", root, "--clip_path", "planted"]
