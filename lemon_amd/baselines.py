"""Scoring API of the reference's kNN-based baselines on the HIP kernels.

  discrepancy_scores   lib/baselines/discrepancy_baseline.py:164-242   dis_x | dis_y | div_x | div_y
  clip_similarity      lib/baselines/run_clip_sim.py:235-248 (DistanceEvaluator.our_metric)
  cos_distance_topk / count_knn_distribution   lib/metrics/utils.py:198-233 (in-tree brute-force kNN)
"""
import ctypes

import torch

from . import _lib
from .index import IndexFlatIP
from .ops import dev_f32, normalize_vectors, our_metric, ptr, stream_ptr


def discrepancy_scores(db, q_img, q_txt, k, method, is_train=False):
    """`db`: LemonDB built with dist_type='cosine' (the baseline only uses IndexFlatIP, :150-155).
    method: 'dis_x' | 'dis_y' | 'div_x' | 'div_y'.  Returns pred_score [nq] (float32, CUDA)."""
    if method not in ("dis_x", "dis_y", "div_x", "div_y"):
        raise NotImplementedError(method)
    assert db.metric == _lib.METRIC_IP, "discrepancy baselines are defined on the cosine (IP) indices"
    q_img, q_txt = dev_f32(q_img, "q_img"), dev_f32(q_txt, "q_txt")
    E = db.img if method.endswith("_x") else db.txt
    qv = q_img if method.endswith("_x") else q_txt
    out = torch.empty(q_txt.shape[0], dtype=torch.float32, device=db.device)
    lib = _lib.load()
    with torch.cuda.device(db.device):
        _lib.check(lib.lemon_discrepancy(0 if method.startswith("dis") else 1, db.index_txt._h, ptr(E), ptr(qv),
                                         ptr(q_txt), q_txt.shape[0], int(k), int(bool(is_train)), ptr(out),
                                         stream_ptr(db.device)), "lemon_discrepancy")
    return out


def clip_similarity(first_modality_embeddings, second_modality_embeddings, dist="cosine"):
    """CLIP-similarity baseline score = paired distance of (un-normalised) image and text embeddings."""
    return our_metric(first_modality_embeddings, second_modality_embeddings, dist)


def clip_logits_confidence(img_embeds, class_text_embeds, noisy_label, dist="cosine"):
    """Zero-shot CLIP-logits baseline (lib/baselines/train_zero_shot_clip_baseline.py:207-224): for every image the
    softmax over ALL class prompts of 1 - our_metric(text_c, image), read at the image's noisy label -> confidence [n]
    (float32 CUDA; low confidence = likely mislabelled).  Embeddings are taken un-normalised, as the baseline does."""
    kind = {"cosine": 0, "euclidean": 1, "manhattan": 2}.get(dist)
    if kind is None:
        raise NotImplementedError(dist)
    q, c = dev_f32(img_embeds, "img_embeds"), dev_f32(class_text_embeds, "class_text_embeds")
    assert q.dim() == 2 and c.dim() == 2 and q.shape[1] == c.shape[1]
    lab = torch.as_tensor(noisy_label).to(device=q.device, dtype=torch.int32).contiguous()
    out = torch.empty(q.shape[0], dtype=torch.float32, device=q.device)
    lib = _lib.load()
    with torch.cuda.device(q.device):
        _lib.check(lib.lemon_class_confidence(kind, ptr(q), q.shape[0], q.shape[1], ptr(c), c.shape[0], ptr(lab), ptr(out),
                                              stream_ptr(q.device)), "lemon_class_confidence")
    return out


def cos_distance_topk(features, k):
    """values, indices of the k smallest cosine distances of every row to all rows (self included), as
    `cosDistance(features).topk(k, largest=False, sorted=True)` (lib/metrics/utils.py:198-212) without
    the N x N matrix: exact flat search on the normalised features."""
    f = normalize_vectors(dev_f32(features, "features"))
    index = IndexFlatIP(f.shape[1], f.device)
    index.add(f)
    D, I = index.search(f, k)
    return 1.0 - D, I


def count_knn_distribution(num_classes, min_similarity, feat_cord, label, k, norm="l2"):
    """lib/metrics/utils.py:205-233: per-sample class distribution of its k nearest neighbours, weighted
    by (1 - min_similarity - distance), with the self-distance patched to 2*v1 - v2 (:214)."""
    values, indices = cos_distance_topk(feat_cord, k)
    values = values.clone()
    values[:, 0] = 2.0 * values[:, 1] - values[:, 2]
    label = torch.as_tensor(label).to(values.device)
    knn_labels = label[indices]
    w = 1.0 - min_similarity - values
    cnt = torch.zeros((values.shape[0], num_classes), dtype=torch.float32, device=values.device)
    cnt.scatter_add_(1, knn_labels.long(), w)
    if norm == "l2":
        return torch.nn.functional.normalize(cnt, p=2.0, dim=1)
    if norm == "l1":
        return cnt / cnt.sum(1, keepdim=True)
    raise NameError("Undefined norm")
