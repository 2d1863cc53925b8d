"""Device-resident restatement of run_lemon.py:121-312 (embed DB -> flat indices -> score every split).

The reference moves every embedding to the CPU (run_lemon.py:158-161,230-233) and loops per sample in
Python (:238-307).  Here embeddings never leave HBM: encoder output -> K1 normalise -> LemonDB
(K2 dists_tr + index.add) -> one lemon_neighbors call per split (K3+K4) -> K5 scores.  With
world_size > 1 every rank embeds its shard of each split, the DB shards are all-gathered over RCCL
(torch.distributed, backend "nccl") and each rank scores its own query shard.
"""
import os
import time

import numpy as np
import torch

from . import ops
from .neighbors import LemonDB

FIXED_HPARAMS = dict(beta=5.0, gamma=5.0, tau_1_n=0.1, tau_2_n=5.0, tau_1_m=0.1, tau_2_m=5.0)
# ^ train_clip_from_scratch.py:102-109 / notebooks/hparam_drop.ipynb cell 4 (SURVEY 8d metric (ii))


def shard_bounds(n, world_size, rank):
    """Contiguous row range of `rank`: rows [r*ceil(n/W), ...)."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class GatherLog:
    """What the path's only exchange step cost ("log the achieved bus bandwidth"): one entry per
    all-gathered array with HIP events recorded around the collective on the caller's stream (the collective's own stream
    is joined by `wait()` before the second event).  `summary()` synchronises the events; call it after the timed region."""

    def __init__(self):
        self.entries = []

    def add(self, name, rows, row_bytes, world, backend, start, end):
        self.entries.append(dict(name=name, rows=int(rows), row_bytes=int(row_bytes), world=int(world), backend=backend,
                                 start=start, end=end))

    def summary(self):
        """{"allgather_ms": total, "arrays": [{name, bytes_gathered, ms, algbw_GBs, busbw_GBs}]}: algbw = gathered bytes /
        time; busbw = algbw x (W-1)/W, the per-rank receive rate a ring or a direct exchange must sustain (nccl-tests
        convention) -- to be read against 7 xGMI links x ~153 GB/s per GPU."""
        out, total = {}, 0.0
        for e in self.entries:
            e["end"].synchronize()
            ms = e["start"].elapsed_time(e["end"])
            a = out.setdefault(e["name"], dict(name=e["name"], calls=0, ms=0.0, bytes_gathered=0, backend=e["backend"]))
            a["calls"] += 1; a["ms"] += ms; a["bytes_gathered"] += e["rows"] * e["row_bytes"]
            total += ms
            W = e["world"]
            a["_w"] = W
        arrays = []
        for a in out.values():
            W = a.pop("_w")
            sec = max(a["ms"], 1e-9) / 1e3
            a["algbw_GBs"] = a["bytes_gathered"] / sec / 1e9
            a["busbw_GBs"] = a["algbw_GBs"] * (W - 1) / W
            arrays.append(a)
        return {"allgather_ms": total, "arrays": arrays}


def all_gather_rows(t, n_total, group=None, log=None, name="rows", force=False):
    """All-gather row shards (possibly ragged: last shard short/empty) into global row order.
    One RCCL all_gather per array; shards are padded to ceil(n/W) rows.  `log`: a GatherLog that
    receives the collective's HIP-event bracket.  A one-rank group returns the shard itself unless `force` (or
    LEMON_FORCE_ALLGATHER=1) sends it through the collective anyway -- how the RCCL path is exercised on a one-GPU box."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if dist.get_world_size(group) == 1 and not (force or os.environ.get("LEMON_FORCE_ALLGATHER") == "1"):
        return t
    W = dist.get_world_size(group)
    per = (n_total + W - 1) // W
    pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    backend = dist.get_backend(group)
    ev = None
    if log is not None and t.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if t.is_cuda and backend == "gloo":
        # rehearsal mode only (several ranks on ONE card, LEMON_DIST_BACKEND=gloo: RCCL needs distinct devices):
        # gloo gathers host buffers, so the shard is staged through the host; the product path is the RCCL branch below
        host = torch.empty((W * per,) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(host, pad.cpu(), group=group)
        out = host[:n_total].to(t.device)
    else:
        out = torch.empty((W * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        out = out[:n_total]
    if ev is not None:
        ev[1].record()
        log.add(name, W * per, t[0].numel() * t.element_size() if t.shape[0] else pad[0].numel() * pad.element_size(), W,
                backend, ev[0], ev[1])
    return out


class Embedder:
    """HOT LOOP 1/2 of run_lemon.py (:137-161, :202-233) without the D2H copies.

    Range fallback.  The default GEMM mode (LEMON_GEMM=f16x3) carries the fp32 operands of the block GEMMs (and q, k, v of
    the attention kernels) as fp16 pairs: a value beyond +-65 504 turns into inf / NaN there instead of being clamped.  Every
    micro-batch leaves a device flag "not finite"; the flags of one embed_images / embed_texts call are read with ONE host
    transfer after all its micro-batches are queued, and the non-finite SAMPLES of a flagged micro-batch (samples are
    independent) are gathered and embedded again with the range-free scheme (3-way bf16 split GEMMs, fp32 attention: the same
    fp32-equivalent contract) -- counted in `fallback_rows` / `fallback_batches`, reported by the CLI and by bench.py's JSON line.  The same path catches the folded LayerNorm's bound (ops.ln_fold_enabled): a row whose mean lies more than
    ops.LN_FOLD_MAX_SHIFT standard deviations from 0 is given a NaN row affine by the kernels, so its micro-batch is flagged
    and its poisoned samples are re-embedded here -- first with the same f16x3 arithmetic and LayerNorm kernels (`fold_fallback_rows`), and only if it
    is still not finite with the range-free scheme.  What is still not finite after that is not a range problem and raises in raise_if_nonfinite()."""

    def __init__(self, model, device, batch_size=128, dtype=torch.float32, text_dedup=False, text_batch_size=None, range_fallback=True,
                 length_bucketing=False, text_token_budget=None):
        self.model = model.eval().to(device=device, dtype=dtype)
        self.device, self.batch_size, self.dtype, self.text_dedup = device, batch_size, dtype, text_dedup
        # prompts are ~7x shorter than the image token sequence: a 4x larger text micro-batch keeps the
        # text tower's GEMMs at the image tower's row count (and efficiency)
        self.text_batch_size = text_batch_size or 4 * batch_size
        self._nonfinite = None        # device flag: some embedding so far was not finite (raise_if_nonfinite)
        self.range_fallback = range_fallback
        self.fallback_batches = 0     # micro-batches that held a sample re-embedded with bf16x6 operands because fp16 overflowed
        self.fold_fallback_batches = 0    # ... with LayerNorm kernels because a row's mean was beyond the fold's bound
        self.fallback_rows = 0        # the samples themselves (only they are embedded again, in sub-batches)
        self.fold_fallback_rows = 0
        # captions (run_lemon.py:140-154 pads every one to the 77-token context): sorted by length before they are cut into
        # micro-batches, so that a micro-batch runs the tokens of ITS longest caption only (exact under the causal mask, like the
        # per-batch truncation without it; a caption's embedding does not depend on its batch mates).  Off by default: on
        # classification prompts all lengths fall into one or two 8-token buckets anyway
        self.length_bucketing = length_bucketing
        # with sorted captions a micro-batch holds `text_token_budget // L` captions of its bucket's length L instead of
        # text_batch_size whatever their length: the GEMMs' row count (and the activations' footprint) stays the same from the
        # 16-token bucket to the 77-token one.  Sized like the image micro-batch's token rows (batch_size x image tokens) the
        # row tiles come out in whole rounds of the chip for every GEMM width: with a fixed caption count the n = width GEMMs of a
        # 512-wide tower (two tile columns) ran 1.3 ... 6.2 rounds by bucket (65 ... 97 % full).  None: text_batch_size captions.
        # Towers that need single-length groups (biomed.BertTextTower) default to the image micro-batch's row count.
        if text_token_budget is None and getattr(getattr(self.model, "text", None), "exact_lengths", False):
            cfg = getattr(self.model, "cfg", None)
            if cfg is not None and hasattr(cfg, "patch_size"):
                text_token_budget = batch_size * ((cfg.image_size // cfg.patch_size) ** 2 + 1)
        self.text_token_budget = text_token_budget
        self.text_tokens_run = 0      # token rows the text tower actually ran (bucketed length x captions), for FLOP accounting

    def _note(self, e):
        if e.is_cuda and e.shape[0]:
            bad = ~torch.isfinite(e).all()
            self._nonfinite = bad if self._nonfinite is None else (self._nonfinite | bad)
        return e

    def raise_if_nonfinite(self):
        """One host read for everything embedded since the last call."""
        flag, self._nonfinite = self._nonfinite, None
        if flag is not None and bool(flag.item()):
            how = ("the range fallback is off (range_fallback=False): an activation left the fp16 range of the split operands (or a row's "
                   "mean left the folded LayerNorm's bound, ops.LN_FOLD_MAX_SHIFT standard deviations) -- "
                   "rerun with LEMON_GEMM=bf16x6 (no range limit) or f32" if (ops.gemm_mode() == "f16x3" and not self.range_fallback)
                   else "also with range-free operands (bf16x6 GEMMs, fp32 attention): the weights or inputs themselves produce inf / NaN")
            raise FloatingPointError(f"non-finite embeddings (LEMON_GEMM={ops.gemm_mode()}): {how}")

    def _run_batches(self, n, bs, run, spans=None):
        """torch.cat of run(slice) over the micro-batches of n samples, with the fallback of the class docstring.  `run(sel)` embeds
        the samples `sel` selects (a slice, or a LongTensor of sample indices).  Samples are independent, so what is embedded again
        is the flagged SAMPLES (gathered into sub-batches of at most bs), not their micro-batches: one poisoned image costs one
        image's work, not 5 240."""
        if spans is None:
            spans = [(i, min(n, i + bs)) for i in range(0, n, bs)]
        outs = [run(slice(lo, hi)).float() for lo, hi in spans]
        if not outs:
            return None
        if not (outs[0].is_cuda and self.range_fallback and ops.gemm_mode() == "f16x3"):
            return torch.cat(outs)
        rowbad = [~torch.isfinite(e).all(dim=1) for e in outs]
        bad = torch.stack([r.any() for r in rowbad]).cpu()                  # the call's one host read
        flagged = bad.nonzero().flatten().tolist()
        e = torch.cat(outs)
        if not flagged:
            return e
        # (second host read, only on this path: which samples)
        idx = torch.cat([spans[j][0] + rowbad[j].nonzero().flatten() for j in flagged]).cpu()

        def redo(rows):
            return torch.cat([run(rows[i:i + bs]).float() for i in range(0, rows.numel(), bs)])

        starts = torch.tensor([lo for lo, _ in spans])

        def batches_of(rows):
            return int(torch.unique(torch.bucketize(rows, starts, right=True)).numel())

        if ops.ln_fold_enabled() and ops.mlp_mode() == "block":
            # first the cheap cause: a row beyond the folded LayerNorm's mean bound (NaN row affine) -- the same arithmetic with
            # LayerNorm kernels; what is still not finite after that (one more host read) left the fp16 range
            with ops.ln_fold_forced(False):
                new = redo(idx)
            ok = torch.isfinite(new).all(dim=1).cpu()
            if bool(ok.any()):
                e[idx[ok].to(e.device)] = new[ok.to(new.device)]
                self.fold_fallback_rows += int(ok.sum())
                self.fold_fallback_batches += batches_of(idx[ok])
            idx = idx[~ok]
        if idx.numel():
            with ops.gemm_mode_forced("bf16x6"):
                e[idx.to(e.device)] = redo(idx)
            self.fallback_rows += int(idx.numel())
            self.fallback_batches += batches_of(idx)
        return e

    @torch.no_grad()
    def embed_images(self, pixel_values):
        """pixel_values: float [n,3,S,S] already preprocessed, or uint8 [n,H,W,3] raw images -- then
        generic_transform runs on the GPU per micro-batch (lemon_preprocess_u8, lib/datasets/utils.py:159-170)."""
        raw = pixel_values.dtype == torch.uint8
        if raw:
            from .data import gpu_transform_batch, patch_operand_supported

        def one(sel):
            px = pixel_values[sel if isinstance(sel, slice) else sel.to(pixel_values.device)].to(self.device, non_blocking=True)
            if raw:
                # f16x3 with the hand-written GEMM: the transform writes the patch-embedding GEMM's operand itself
                cfg = self.model.cfg
                operand = (ops.gemm_mode() == "f16x3" and ops.mlp_mode() != "lib" and patch_operand_supported(cfg.patch_size, cfg.image_size)
                           and cfg.vision.width % 256 == 0 and hasattr(self.model, "vision")
                           and os.environ.get("LEMON_PATCH_OPERAND", "1") != "0")          # (=0: A/B knob, the fp32 patch rows + split pass)
                px = gpu_transform_batch(px, cfg.image_size, patch=cfg.patch_size, operand=operand)
            return self.model.encode_image(px)

        e = self._run_batches(pixel_values.shape[0], self.batch_size, one)
        if e is None:
            e = torch.empty((0, self.model.cfg.embed_dim), device=self.device)
        return ops.normalize_vectors(self._note(e)) if e.shape[0] else e              # :164 / :233

    @torch.no_grad()
    def embed_texts(self, input_ids):
        eot = None if input_ids.is_cuda else self._last_token(input_ids)    # host ids: EOT positions without a sync
        input_ids = input_ids.to(self.device)
        if self.text_dedup:
            uniq, inv = torch.unique(input_ids, dim=0, return_inverse=True)
            e = self._embed_texts(uniq)[inv]
        else:
            e = self._embed_texts(input_ids, eot)
        return ops.normalize_vectors(self._note(e)) if e.shape[0] else e              # :163 / :230-232

    def _last_token(self, ids):
        """per-row index of the caption's last token: the EOT of the CLIP towers (the largest id, chexzero_clip.py:374-376), the
        last non-pad token of a BERT tower (lemon_amd/biomed.py)"""
        tower = getattr(self.model, "text", None)
        return tower.last_token_index(ids) if hasattr(tower, "last_token_index") else ids.argmax(dim=-1)

    def _embed_texts(self, ids, eot=None):
        """ids on the device; eot = per-row EOT position on the HOST (one transfer for the whole array instead of a
        device sync per micro-batch) from which each micro-batch's bucketed token count is taken."""
        if ids.shape[0] == 0:
            return torch.empty((0, self.model.cfg.embed_dim), device=self.device)
        tower = getattr(self.model, "text", None)
        if eot is None and hasattr(tower, "seq_len_for"):
            eot = self._last_token(ids).cpu()
        bucketed = eot is not None and hasattr(tower, "seq_len_for")
        # a tower without a padding mask in its kernels (BERT: bidirectional attention) runs every caption at exactly its own
        # length: its micro-batches must be single-length groups, so its captions are always sorted
        exact = bucketed and getattr(tower, "exact_lengths", False)

        perm = spans = None
        if bucketed and (exact or (self.length_bucketing and ids.shape[0] > self.text_batch_size)):
            perm = torch.argsort(eot, stable=True)           # host: shortest captions first
            # micro-batches never straddle a token bucket: every caption runs exactly its own bucket's tokens
            L_sorted = torch.tensor([tower.seq_len_for(int(v)) for v in eot[perm].tolist()])
            spans, lo = [], 0
            for Lb, cnt in zip(*[t.tolist() for t in torch.unique_consecutive(L_sorted, return_counts=True)]):
                per = self.text_batch_size if not self.text_token_budget else max(1, int(self.text_token_budget) // max(1, int(Lb)))
                for i in range(lo, lo + cnt, per):
                    spans.append((i, min(lo + cnt, i + per)))
                lo += cnt

        def one(sel):
            if perm is not None:                             # sel addresses the length-sorted order
                sel = perm[sel]
            rows = ids[sel if isinstance(sel, slice) else sel.to(ids.device)]
            if bucketed:
                L = tower.seq_len_for(int(eot[sel].max()))
                self.text_tokens_run += int(L) * int(rows.shape[0])
                if exact:
                    return self.model.encode_text(rows, seq_len=L, lengths=eot[sel] + 1)
                return self.model.encode_text(rows, seq_len=L)
            return self.model.encode_text(rows)

        e = self._run_batches(ids.shape[0], self.text_batch_size, one, spans=spans)
        if perm is not None:
            out = torch.empty_like(e)
            out[perm.to(e.device)] = e
            e = out
        return e


def score_splits(db, splits, k, hparams=None, discrete=False):
    """splits: list of dicts {name, img, txt, drop_self, in_db, label_id}.  Returns per split the
    record of device arrays (+ 'score' float64 when hparams is given).

    All splits go through ONE lemon_neighbors call (one scan launch per modality instead of one per
    split and modality): the train split searches k+1 and drops result[0] where the sample is in the DB
    else result[-1] (run_lemon.py:257-263); val/test search k.  With the per-query `in_db` flag the
    second case is exactly "keep the first k of k+1", so val/test ride along with in_db = 0."""
    splits = [s for s in splits if s["img"].shape[0] > 0]
    if not splits:
        return {}
    any_train = any(s.get("drop_self", False) for s in splits)
    dev = splits[0]["img"].device
    img = torch.cat([s["img"] for s in splits])
    txt = torch.cat([s["txt"] for s in splits])
    in_db = None
    if any_train:
        parts = []
        for s in splits:
            n = s["img"].shape[0]
            if s.get("drop_self", False):
                m = s.get("in_db")
                parts.append(torch.ones(n, dtype=torch.uint8, device=dev) if m is None
                             else torch.as_tensor(m).to(device=dev, dtype=torch.uint8))
            else:
                parts.append(torch.zeros(n, dtype=torch.uint8, device=dev))
        in_db = torch.cat(parts)
    lab = None
    if discrete:
        lab = torch.cat([torch.as_tensor(s["label_id"]).to(device=dev, dtype=torch.int32) for s in splits])
    rec = db.neighbors(img, txt, k, drop_self=any_train, in_db=in_db, discrete=discrete, q_label_id=lab)
    if os.environ.get("LEMON_DEBUG_FINITE"):       # diagnostic: name the first non-finite array (host sync)
        for key, v in (("emb_img", img), ("emb_txt", txt)) + tuple(rec.items()):
            if v.is_floating_point() and not bool(torch.isfinite(v).all()):
                bad = (~torch.isfinite(v)).nonzero()[:5].tolist()
                raise FloatingPointError(f"non-finite values in {key} at {bad} (k={k}, ntotal={db.index_img.ntotal})")
    if hparams is not None:
        rec["score"] = ops.lemon_score(rec, hparams)
    out, lo = {}, 0
    for s in splits:
        n = s["img"].shape[0]
        out[s["name"]] = {key: v[lo:lo + n] for key, v in rec.items()}
        lo += n
    return out


def run_hot_path(embedder, data, k=5, dist_type="cosine", hparams=FIXED_HPARAMS, discrete=False,
                 world_size=1, rank=0, algo=None, timers=None, profile_index=False, gather_log=None):
    """One pass of the hot path over `data` = {split: {"pixels": float [n,3,S,S] or raw uint8 [n,H,W,3], "ids": [n,L], "label_id": [n]}}
    for split in train/val/test, this rank's shard of each.  DB = all ranks' train shards in global
    order.  The train split is embedded ONCE and reused as DB and as queries (the reference embeds it
    twice, run_lemon.py:137-161 and :198-233; same model, same inputs => same embeddings)."""
    dev = embedder.device
    t0 = time.perf_counter()
    emb = {}
    for name, d in data.items():
        emb[name] = (embedder.embed_images(d["pixels"]), embedder.embed_texts(d["ids"]))
    embedder.raise_if_nonfinite()
    if timers is not None:
        torch.cuda.synchronize(dev)
        timers["embed_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
    n_train_total = data["train"].get("n_total", emb["train"][0].shape[0])
    img_tr = all_gather_rows(emb["train"][0], n_train_total, log=gather_log, name="emb_img_tr")
    txt_tr = all_gather_rows(emb["train"][1], n_train_total, log=gather_log, name="emb_txt_tr")
    lab = data["train"].get("label_id")
    lab_tr = all_gather_rows(lab.to(dev), n_train_total, log=gather_log, name="label_id_tr") if (discrete and lab is not None) else None
    db_index = data["train"].get("db_index")
    if db_index is not None:                # DB = a subset of the train split, in the order drawn (run_lemon.py:121-127); the train
        sel = torch.as_tensor(db_index).to(dev)     # queries then carry data["train"]["in_db"] (self-exclusion, :256-263)
        img_tr, txt_tr = img_tr[sel].contiguous(), txt_tr[sel].contiguous()
        lab_tr = lab_tr[sel].contiguous() if lab_tr is not None else None
    db = LemonDB(img_tr, txt_tr, dist_type, tr_label_id=lab_tr, algo=algo)
    if profile_index:
        db.index_img.set_profiling(True)
        db.index_txt.set_profiling(True)
    splits = []
    for name in ("train", "val", "test"):
        if name not in data:
            continue
        splits.append(dict(name=name, img=emb[name][0], txt=emb[name][1], drop_self=(name == "train"),
                           in_db=data[name].get("in_db"), label_id=data[name].get("label_id")))
    recs = score_splits(db, splits, k, hparams, discrete)
    if timers is not None:
        torch.cuda.synchronize(dev)
        timers["knn_score_s"] = time.perf_counter() - t0
    for name in recs:                       # keep the embeddings the scores were computed from
        recs[name]["emb_img"], recs[name]["emb_txt"] = emb[name]
    return recs, db
