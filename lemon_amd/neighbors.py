"""The per-sample loop of run_lemon.py:238-307 as one device call per split."""
import ctypes

import torch

from . import _lib
from .index import IndexFlatIP, IndexFlatL2
from .ops import dev_f32, metric_id, paired_distance, ptr, stream_ptr

REC_KEYS = ("d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m", "I_n", "I_m")


class LemonDB:
    """DB side of run_lemon.py:163-176: normalised train embeddings, the two flat indices and
    dists_tr.  Everything stays in HBM."""

    def __init__(self, emb_img_tr, emb_txt_tr, dist_type="cosine", tr_label_id=None, algo=None):
        self.metric = metric_id(dist_type)
        self.img = dev_f32(emb_img_tr, "emb_img_tr")
        self.txt = dev_f32(emb_txt_tr, "emb_txt_tr")
        assert self.img.shape == self.txt.shape and self.img.dim() == 2
        self.device = self.img.device
        cls = IndexFlatIP if self.metric == _lib.METRIC_IP else IndexFlatL2
        d = self.img.shape[1]
        with torch.cuda.device(self.device):
            self.index_img, self.index_txt = cls(d, self.device), cls(d, self.device)   # :167-168 / :171-172
            if algo is not None:
                self.index_img.set_algo(algo)
                self.index_txt.set_algo(algo)
            self.dists_tr = paired_distance(self.metric, self.txt, self.img)            # :169 / :173
            self.index_txt.add(self.txt)                                                # :175
            self.index_img.add(self.img)                                                # :176
        self.tr_label_id = None if tr_label_id is None else \
            torch.as_tensor(tr_label_id).to(device=self.device, dtype=torch.int32).contiguous()

    @property
    def ntotal(self):
        return self.img.shape[0]

    def neighbors(self, q_img, q_txt, k, drop_self=False, in_db=None, discrete=False, q_label_id=None,
                  return_indices=True):
        """One split of the scoring loop.  Returns a dict of CUDA tensors keyed like the reference's
        per-sample record (run_lemon.py:291-307) plus I_n / I_m."""
        q_img, q_txt = dev_f32(q_img, "q_img"), dev_f32(q_txt, "q_txt")
        nq, d = q_img.shape
        assert q_txt.shape == (nq, d) and d == self.img.shape[1]
        dev = self.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = {"d_1": f(nq)}
        for nm in ("D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
            out[nm] = f(nq, k)
        if return_indices:
            out["I_n"] = torch.empty((nq, k), dtype=torch.int64, device=dev)
            out["I_m"] = torch.empty((nq, k), dtype=torch.int64, device=dev)
        in_db_t = None
        if drop_self and in_db is not None:
            in_db_t = torch.as_tensor(in_db).to(device=dev, dtype=torch.uint8).contiguous()
        q_lab = None
        if discrete:
            if self.tr_label_id is None or q_label_id is None:
                raise ValueError("use_discrete_for_text needs tr_label_id and q_label_id")
            q_lab = torch.as_tensor(q_label_id).to(device=dev, dtype=torch.int32).contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.lemon_neighbors(
                self.index_img._h, self.index_txt._h, ptr(self.dists_tr), ptr(q_img), ptr(q_txt), nq, int(k),
                int(bool(drop_self)), ptr(in_db_t), int(bool(discrete)), ptr(self.tr_label_id), ptr(q_lab),
                ptr(out["d_1"]), ptr(out["D_n"]), ptr(out["dists_n"]), ptr(out["dists_tr_n"]), ptr(out.get("I_n")),
                ptr(out["D_m"]), ptr(out["dists_m"]), ptr(out["dists_tr_m"]), ptr(out.get("I_m")),
                stream_ptr(dev)), "lemon_neighbors")
        return out
