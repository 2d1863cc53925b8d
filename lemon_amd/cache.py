"""On-disk embedding cache for run_lemon: the normalised image / text embeddings of one split
shard, so the kNN + scoring stage can be re-run (other k, metric, ablation, hyper-parameters) without the encoder --
the reference re-embeds everything on every invocation, and the train split twice (run_lemon.py:137-161,198-233).

One entry = `<dir>/<key>/{img.npy, txt.npy, meta.pkl}`; the key hashes everything the embeddings depend on (dataset,
noise, seeds, checkpoint, prompt prefix, split, shard) and NOT what they do not (k, metric, ablation, search flags)."""
import hashlib
import json
import os
import pickle

import numpy as np
import torch


class EmbeddingCache:
    def __init__(self, root, **fields):
        self.root = root
        self.fields = {k: (str(v) if v is not None else None) for k, v in sorted(fields.items())}
        if root:
            os.makedirs(root, exist_ok=True)

    def _dir(self, split, lo, hi, prompts):
        h = hashlib.sha1(json.dumps([self.fields, split, lo, hi]).encode())
        for p in prompts:                                  # the texts actually embedded (labels + noise realisation)
            h.update(b"\0" + str(p).encode("utf-8"))
        return os.path.join(self.root, h.hexdigest()[:24])

    def load(self, split, lo, hi, prompts, device):
        if not self.root:
            return None
        d = self._dir(split, lo, hi, prompts)
        if not os.path.exists(os.path.join(d, "done")):
            return None
        img = torch.from_numpy(np.load(os.path.join(d, "img.npy"))).to(device)
        txt = torch.from_numpy(np.load(os.path.join(d, "txt.npy"))).to(device)
        with open(os.path.join(d, "meta.pkl"), "rb") as f:
            meta = pickle.load(f)
        return img, txt, meta

    def store(self, split, lo, hi, prompts, img, txt, meta):
        if not self.root:
            return
        d = self._dir(split, lo, hi, prompts)
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "img.npy"), img.detach().cpu().numpy())
        np.save(os.path.join(d, "txt.npy"), txt.detach().cpu().numpy())
        with open(os.path.join(d, "meta.pkl"), "wb") as f:
            pickle.dump(meta, f)
        with open(os.path.join(d, "done"), "w") as f:      # written last: a partial entry is never read
            f.write("done")
