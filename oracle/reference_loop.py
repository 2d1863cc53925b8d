"""CPU restatement of the reference's timed region in the reference's own terms (torch on the host, a flat index searched
in 128-query batches, a per-sample Python loop) -- the baseline BASELINE.md section 3 / SURVEY 8d define.

TEST INFRASTRUCTURE ONLY (oracle/): used by tests/ and by bench.py's cpu_baseline leg, never by lemon_amd/.

  FlatIndexTorch          stand-in for faiss.IndexFlatIP / IndexFlatL2 as run_lemon.py:166-176,235-236 uses them: float32
                          torch.mm of the query batch against the whole DB + exact top-k, best first (faiss itself is not
                          installable here and un-vendored upstream: SURVEY 8c)
  per_sample_loop         run_lemon.py:238-307 line by line: d_1 (:243-253), the self-exclusion with the O(N) membership test
                          `sample_idx in train_indices_in_compr` on a numpy array (:256-263, :276-283), the fancy-index gathers
                          and cross-modal distances (:264-273, :284-289), the sign quirk (:269-270, :285-286), one dict per
                          sample (:291-307)
  vectorised              the same arithmetic for a whole split in numpy (what a CPU user would write instead of the loop)
  adjudicate_near_ties    for the queries on which two float32 searches return different neighbour SETS: the float64 scores of
                          the rows in dispute, which side holds the float64-exact set, and how far apart the swapped rows are
"""
import numpy as np
import torch


class FlatIndexTorch:
    def __init__(self, d, metric):
        assert metric in ("cosine", "euclidean")
        self.d, self.metric, self.x = d, metric, None

    def add(self, x):                                   # run_lemon.py:175-176 (numpy float32 [N, d]; copied)
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).clone()
        assert x.shape[1] == self.d
        self.x = x if self.x is None else torch.cat([self.x, x])
        self.xn = (self.x * self.x).sum(1)

    def search(self, q, k):                             # run_lemon.py:235-236 -> (D float32 [nq, k], I int64 [nq, k])
        q = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32))
        s = q @ self.x.t()
        if self.metric == "cosine":
            D, I = torch.topk(s, k, dim=1, largest=True, sorted=True)
        else:                                           # squared L2 by the expansion, smallest first
            d2 = (q * q).sum(1, keepdim=True) + self.xn[None, :] - 2.0 * s
            D, I = torch.topk(d2, k, dim=1, largest=False, sorted=True)
        return D.numpy(), I.numpy()


def per_sample_loop(sname, img_embeds_all, text_embeds_all, emb_img_tr, emb_txt_tr, dists_tr, index_img, index_txt, k, bs,
                    train_indices_in_compr, dist_type="cosine"):
    """img_embeds_all / text_embeds_all: normalised CPU torch tensors [n, d] of one split; emb_*_tr torch [N, d]; dists_tr torch
    [N]; train_indices_in_compr numpy int array (the DB's rows of the train split).  Returns the list of per-sample dicts."""
    logs = []
    n = img_embeds_all.shape[0]
    for idx in range((n + bs - 1) // bs):
        img_embeds = img_embeds_all[idx * bs:(idx + 1) * bs]
        text_embeds = text_embeds_all[idx * bs:(idx + 1) * bs]
        D_ns, I_ns = index_img.search(img_embeds.numpy(), k + (sname == 'train'))
        D_ms, I_ms = index_txt.search(text_embeds.numpy(), k + (sname == 'train'))
        for i in range(len(img_embeds)):
            sample_idx = idx * bs + i
            img_embed = img_embeds[i, None]
            text_embed = text_embeds[i, None]
            if dist_type == 'cosine':
                d1 = 1 - torch.dot(img_embed.flatten(), text_embed.flatten())
            else:
                d1 = ((img_embed.flatten() - text_embed.flatten()) ** 2).sum()
            D_n, I_n = D_ns[i], I_ns[i]
            if sname == 'train':
                if sample_idx in train_indices_in_compr:          # O(N) scan of a numpy array, as upstream
                    I_n = I_n[1:]; D_n = D_n[1:]
                else:
                    I_n = I_n[:-1]; D_n = D_n[:-1]
            y_n = emb_txt_tr[I_n]
            if dist_type == 'cosine':
                D_n = -D_n
                dists_n = 1 - (text_embed * y_n).sum(axis=1)
            else:
                dists_n = ((text_embed - y_n) ** 2).sum(axis=1)
            D_m, I_m = D_ms[i], I_ms[i]
            if sname == 'train':
                if sample_idx in train_indices_in_compr:
                    I_m = I_m[1:]; D_m = D_m[1:]
                else:
                    I_m = I_m[:-1]; D_m = D_m[:-1]
            x_m = emb_img_tr[I_m]
            if dist_type == 'cosine':
                D_m = -D_m
                dists_m = 1 - (img_embed * x_m).sum(axis=1)
            else:
                dists_m = ((img_embed - x_m) ** 2).sum(axis=1)
            logs.append({'sset': sname, 'idx': sample_idx, 'd_1': d1.item(), 'dists_n': dists_n.numpy(), 'D_n': D_n.flatten(),
                         'dists_tr_n': dists_tr[I_n].numpy(), 'dists_m': dists_m.numpy(), 'D_m': D_m.flatten(),
                         'dists_tr_m': dists_tr[I_m].numpy(), 'I_n': I_n, 'I_m': I_m})
    return logs


def vectorised(sname, img_q, txt_q, emb_img_tr, emb_txt_tr, dists_tr, index_img, index_txt, k, bs, in_db, dist_type="cosine"):
    """The same quantities for a whole split with numpy array operations (searches still in `bs`-query batches).
    in_db: bool [n] (the sample is a DB row) for the train split, ignored otherwise.  Returns a dict of arrays."""
    img_q = np.asarray(img_q, dtype=np.float32); txt_q = np.asarray(txt_q, dtype=np.float32)
    img_tr = np.asarray(emb_img_tr, dtype=np.float32); txt_tr = np.asarray(emb_txt_tr, dtype=np.float32)
    dtr = np.asarray(dists_tr, dtype=np.float32)
    n = img_q.shape[0]
    kk = k + (sname == 'train')
    Dn = np.empty((n, kk), np.float32); In = np.empty((n, kk), np.int64)
    Dm = np.empty((n, kk), np.float32); Im = np.empty((n, kk), np.int64)
    for lo in range(0, n, bs):
        Dn[lo:lo + bs], In[lo:lo + bs] = index_img.search(img_q[lo:lo + bs], kk)
        Dm[lo:lo + bs], Im[lo:lo + bs] = index_txt.search(txt_q[lo:lo + bs], kk)
    if sname == 'train':
        first = np.asarray(in_db, dtype=bool)[:, None]
        cols = np.arange(k)[None, :] + first                      # drop result[0] where the sample is in the DB, else result[-1]
        rows = np.arange(n)[:, None]
        Dn, In, Dm, Im = Dn[rows, cols], In[rows, cols], Dm[rows, cols], Im[rows, cols]
    if dist_type == 'cosine':
        d1 = 1 - (img_q * txt_q).sum(1)
        dists_n = 1 - np.einsum('nd,nkd->nk', txt_q, txt_tr[In])
        dists_m = 1 - np.einsum('nd,nkd->nk', img_q, img_tr[Im])
        Dn, Dm = -Dn, -Dm
    else:
        d1 = ((img_q - txt_q) ** 2).sum(1)
        dists_n = ((txt_q[:, None, :] - txt_tr[In]) ** 2).sum(2)
        dists_m = ((img_q[:, None, :] - img_tr[Im]) ** 2).sum(2)
    return {'d_1': d1, 'D_n': Dn, 'dists_n': dists_n, 'dists_tr_n': dtr[In], 'I_n': In,
            'D_m': Dm, 'dists_m': dists_m, 'dists_tr_m': dtr[Im], 'I_m': Im}


def stack(logs, key):
    return np.stack([np.asarray(l[key]) for l in logs])


def adjudicate_near_ties(q, x, I_a, I_b, metric="cosine", exclude=None):
    """Two exact float32 searches of the same queries (I_a, I_b: int [nq, k] neighbour indices, any order) can disagree where
    the k-th and (k+1)-th scores are closer than float32 summation noise (hard part 2: the GPU scan sums in chain
    order, torch.mm in blocked order).  For every query whose two SETS differ this computes, in float64 over the WHOLE database,
    the exact top-k set (ties to the lower index; `exclude[i]` = a row to leave out, the self match of a train query, or -1) and
      gap = max - min float64 score over the symmetric difference of the two sets: how far apart the rows in dispute are.
    Returns {rows, rows_differing, rows_a_equals_f64_set, rows_b_equals_f64_set (both over ALL rows, identical rows counted as
    agreeing with float64 only if they do), max_gap_at_swap, worst_row}."""
    q = np.asarray(q, dtype=np.float64); x = np.asarray(x, dtype=np.float64)
    Sa, Sb = np.sort(np.asarray(I_a), 1), np.sort(np.asarray(I_b), 1)
    nq, k = Sa.shape
    differ = np.nonzero((Sa != Sb).any(1))[0]
    a_ok = b_ok = nq - len(differ)          # rows on which both agree: checked against float64 below as well
    max_gap, worst = 0.0, -1
    check = list(differ)
    # a sample of the agreeing rows too (all of them when there are few): agreement of two float32 searches is not proof
    agree = np.setdiff1d(np.arange(nq), differ)
    check_agree = agree[:: max(1, len(agree) // 256)]
    for i in list(differ) + list(check_agree):
        s = x @ q[i] if metric == "cosine" else -((x - q[i][None, :]) ** 2).sum(1)
        if exclude is not None and exclude[i] >= 0:
            s[exclude[i]] = -np.inf
        order = np.lexsort((np.arange(len(s)), -s))[:k]          # score descending, index ascending
        exact = np.sort(order)
        if i in check_agree and not (Sa[i] != Sb[i]).any():
            if not np.array_equal(exact, Sa[i]):
                a_ok -= 1; b_ok -= 1
            continue
        a_ok += int(np.array_equal(exact, Sa[i])); b_ok += int(np.array_equal(exact, Sb[i]))
        sym = np.setxor1d(Sa[i], Sb[i])
        gap = float(s[sym].max() - s[sym].min())
        if gap > max_gap:
            max_gap, worst = gap, int(i)
    return {"rows": int(nq), "rows_differing": int(len(differ)), "rows_a_equals_f64_set": a_ok / nq, "rows_b_equals_f64_set": b_ok / nq,
            "max_gap_at_swap": max_gap, "worst_row": worst, "agreeing_rows_checked_against_f64": int(len(check_agree))}
