"""python -m lemon_amd.run_lemon -- drop-in for the reference's `python -m run_lemon` (run_lemon.py).

Same 21 flags, defaults and choices (run_lemon.py:35-57), same outputs (args.json, out.txt, err.txt,
res.pkl {'df','agg_results'}, know_val_labels_scores.csv, need_hparam_optim, done) and the same
per-sample record schema (:291-307).  Differences, all explicit:
  * embeddings stay in HBM; kNN + per-sample quantities + scores run in liblemon_hip.so (no faiss, no
    per-sample Python loop); the train split is embedded once and reused as DB and as queries;
  * weights/data come from LOCAL paths (extension flags --clip_path, --data_root, --arch, --algo,
    --encoder_batch); `--clip_path random[:arch]` / `--data_root synthetic:N` run without any files;
  * under torchrun (WORLD_SIZE>1) samples are sharded over ranks, DB shards all-gathered over RCCL.
"""
import argparse
import json
import os
import pickle
import random
import socket
import sys
from datetime import datetime
from pathlib import Path

import numpy as np
import pandas as pd
import torch

CLF_DATASETS = ["cifar10", "cifar100", "cifar10_full", "cifar100_full", "mini_imagenet", "stanford_cars"]


def build_parser():
    p = argparse.ArgumentParser(description="LEMoN")
    p.add_argument("--exp_name", type=str)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--dataset", type=str, default="cifar100",
                   choices=["cifar10", "cifar100", "flickr30k", "mscoco", "mimiccxr_caption", "mmimdb", "cifar10_full",
                            "cifar100_full", "mini_imagenet", "stanford_cars", "cc3m"])
    p.add_argument("--noise_type", type=str, default="real",
                   choices=["real", "asymmetric", "symmetric", "random", "noun", "cat"])
    p.add_argument("--noise_level", type=float, default=0.4)
    p.add_argument("--dist_type", type=str, default="cosine", choices=["cosine", "euclidean"])
    p.add_argument("--normalize_d1", action="store_true",
                   help="normalize CLIP sim by all possible labels. Only for CIFAR-10 and CIFAR-100")
    p.add_argument("--clip_model", type=str, default="huggingface_clip",
                   choices=["huggingface_clip", "biomed_clip", "mimic_clip_from_scratch_random",
                            "mimic_clip_from_scratch_cat", "chexzero", "cc3m_clip_from_scratch"])
    p.add_argument("--knn_k", default=5, type=int)
    p.add_argument("--batch_size", default=128, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--data_seed", default=0, type=int)
    p.add_argument("--compr_dataset_size_limit", default=50000, type=int)
    p.add_argument("--ablation", default="none",
                   choices=["none", "tau_1", "tau_2", "tau_1_2", "beta", "gamma", "multimodal_baseline", "d1",
                            "only_gamma", "only_beta"])
    p.add_argument("--use_discrete_for_text", action="store_true", help="use the discrete metric for text comparisons")
    p.add_argument("--real_dataset", action="store_true", help="Running on real dataset, do not optimize hparams")
    p.add_argument("--custom_cifar_prompt", default=None)
    p.add_argument("--subset_val_set", default=-1, type=int)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--skip_train", action="store_true")
    p.add_argument("--skip_hparam_optim", action="store_true")
    # ---- extensions (not in the reference) ----
    p.add_argument("--clip_path", default="random",
                   help="local HF CLIP checkpoint dir (huggingface_clip), local OpenAI-format .pt (in-tree CLIP branches), or random[:arch]")
    p.add_argument("--bpe_path", default=None,
                   help="CLIP BPE merges file (bpe_simple_vocab_16e6.txt.gz / merges.txt) when the checkpoint has no tokenizer files")
    p.add_argument("--data_root", default="./data", help="local dataset root, or synthetic:N")
    p.add_argument("--algo", default="auto", choices=["auto", "f32", "bf16"], help="kNN scan algorithm")
    p.add_argument("--encoder_batch", default=512, type=int, help="encoder micro-batch on the GPU")
    p.add_argument("--lbfgs_device", default="cuda", choices=["cuda", "cpu"],
                   help="where the SoftMargin/LBFGS polish of the hyper-parameter search runs (lib/metrics/utils.py:121-149)")
    p.add_argument("--no_text_dedup", action="store_true",
                   help="encode every sample's prompt (the reference does) instead of each distinct prompt once")
    p.add_argument("--embedding_cache", default=None,
                   help="directory for per-split embedding caches (lemon_amd/cache.py): re-runs with another k / metric / "
                        "ablation skip the encoder")
    p.add_argument("--hparam_grid", default="full", choices=["full", "small"],
                   help="'small' = 3x3x2x2 grid for smoke runs (reference grid is 21x21x4x4)")
    return p


class Tee:
    """lib/utils/utils.py:42-54"""

    def __init__(self, fname, stream, mode="a"):
        self.stream, self.file = stream, open(fname, mode)

    def write(self, m):
        self.stream.write(m); self.file.write(m); self.flush()

    def flush(self):
        self.stream.flush(); self.file.flush()


def main(argv=None):
    args = build_parser().parse_args(argv)
    saved = (sys.stdout, sys.stderr)
    try:
        return _run(args)
    finally:                      # the Tee objects are per run (the reference is a one-shot script; this is a function)
        for cur, old in ((sys.stdout, saved[0]), (sys.stderr, saved[1])):
            if isinstance(cur, Tee):
                cur.file.close()
        sys.stdout, sys.stderr = saved


def _run(args):
    hparams = vars(args)
    out_dir = Path(args.output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.debug and rank == 0:
        sys.stdout = Tee(os.path.join(args.output_dir, "out.txt"), sys.stdout)
        sys.stderr = Tee(os.path.join(args.output_dir, "err.txt"), sys.stderr)

    from . import _lib, datasets as ds, metrics as M, ops
    from .clip import algorithm_class_from_scratch
    from .data import get_dataset
    from .neighbors import LemonDB
    from .pipeline import Embedder, all_gather_rows, shard_bounds

    _lib.load()
    if not torch.cuda.is_available():
        raise _lib.LemonHipError("run_lemon needs a HIP device (no CPU fallback for the hot path)")
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))     # rehearsals: ranks may share one card
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LEMON_DIST_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only to rehearse on one card
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if rank == 0:
        print("Environment:")
        print("\tPython: {}".format(sys.version.split(" ")[0]))
        print("\tPyTorch: {}".format(torch.__version__))
        print("\tHIP: {}".format(torch.version.hip))
        print("\tNumPy: {}".format(np.__version__))
        print("\tNode: {}".format(socket.gethostname()))
        print("\tDevice: {} x{}".format(torch.cuda.get_device_name(device), world))
        print("Args:")
        for k, v in sorted(hparams.items()):
            print("\t{}: {}".format(k, v))

    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if rank == 0:
        with open(out_dir / "args.json", "w") as f:
            json.dump(vars(args), f, default=str)
    if args.real_dataset:
        assert args.noise_level == 0.0

    label_set = ds.LABEL_SETS.get(args.dataset)
    is_clf = args.dataset in CLF_DATASETS
    model, tokenizer = algorithm_class_from_scratch(args.clip_model, text_base_name=args.clip_path, img_base=None,
                                                    return_tokenizer=True, bpe_path=args.bpe_path)
    hf_style = args.clip_model == "huggingface_clip"      # the other branches' tokenizer returns a LongTensor (:148-154)
    train_set, val_set, test_set = get_dataset(args.dataset, args.data_seed, percent_flips=args.noise_level,
                                               flip_type=args.noise_type, data_root=args.data_root,
                                               image_size=model.cfg.image_size)
    if args.subset_val_set > 0:
        rng = np.random.default_rng(args.data_seed)
        val_set = val_set.subset(rng.choice(np.arange(len(val_set)), min(args.subset_val_set, len(val_set)), replace=False))

    # identical prompts are embedded once and gathered (class datasets: C prompts for N samples): the same function
    # of the input, independent of how samples fall into micro-batches; --no_text_dedup encodes every sample's prompt
    embedder = Embedder(model, device, batch_size=args.encoder_batch, text_dedup=not args.no_text_dedup)
    prefix = "A photo of a " if args.custom_cifar_prompt is None else args.custom_cifar_prompt
    prompt_fn = lambda x: prefix + x

    # DB subset: first consumer of the global numpy stream after seeding (run_lemon.py:81,121-127)
    if len(train_set) > args.compr_dataset_size_limit:
        train_indices_in_compr = np.random.choice(np.arange(len(train_set)), args.compr_dataset_size_limit, replace=False)
    else:
        train_indices_in_compr = np.arange(len(train_set))

    def texts_of(noisy, clean):
        if is_clf:
            noisy_txt = label_set[np.asarray(noisy)].tolist()
            clean_txt = label_set[np.asarray(clean)].tolist()
            return noisy_txt, clean_txt, [prompt_fn(t) for t in noisy_txt]
        return list(noisy), list(clean), list(noisy)

    def tokenize(prompts):
        if not hf_style:
            return tokenizer(prompts)
        enc = tokenizer(prompts, padding="max_length", truncation=True)
        return torch.tensor(enc["input_ids"])

    # int ids for the discrete text metric (it compares prompt STRINGS, :266-267).  Class datasets: the
    # prompt is a bijection of the noisy label.  Captions: a dictionary built from the dataset itself,
    # in a fixed order, so every rank derives the same ids.
    text_ids = {}
    if not is_clf:
        for dset in (train_set, val_set, test_set):
            for cap in dset.noisy:
                text_ids.setdefault(cap, len(text_ids))

    def ids_of(meta):
        if is_clf:
            return np.asarray(meta["noisy"], dtype=np.int32)
        return np.array([text_ids[p] for p in meta["prompts"]], dtype=np.int32)

    from .cache import EmbeddingCache
    cache = EmbeddingCache(args.embedding_cache, dataset=args.dataset, noise_type=args.noise_type,
                           noise_level=args.noise_level, data_seed=args.data_seed, clip_model=args.clip_model,
                           clip_path=os.path.abspath(args.clip_path) if os.path.exists(str(args.clip_path)) else args.clip_path,
                           data_root=args.data_root, prompt=args.custom_cifar_prompt, subset_val_set=args.subset_val_set)

    def embed_split(dset, sname):
        """this rank's contiguous shard of a split -> (emb_img, emb_txt, meta) on the device"""
        lo, hi = shard_bounds(len(dset), world, rank)
        if cache.root:
            sl = slice(lo, hi)
            key_prompts = texts_of(dset.noisy[sl], dset.clean[sl])[2]
            hit = cache.load(sname, lo, hi, key_prompts, device)
            if hit is not None:
                return hit
        imgs, toks, meta = [], [], dict(noisy=[], clean=[], noisy_txt=[], clean_txt=[], prompts=[])
        # data chunks of the encoder micro-batch (the reference's --batch_size only sizes its DataLoader batches;
        # per-sample results do not depend on it)
        for px, clean, noisy in dset.batches(max(args.batch_size, args.encoder_batch), lo, hi, device=device):
            noisy_txt, clean_txt, prompts = texts_of(noisy, clean)
            imgs.append(embedder.embed_images(px))
            toks.append(tokenize(prompts))
            meta["noisy"] += list(noisy); meta["clean"] += list(clean)
            meta["noisy_txt"] += noisy_txt; meta["clean_txt"] += clean_txt; meta["prompts"] += prompts
        d = embedder.model.cfg.embed_dim
        e_img = torch.cat(imgs) if imgs else torch.empty((0, d), device=device)
        e_txt = embedder.embed_texts(torch.cat(toks)) if toks else torch.empty((0, d), device=device)
        meta["lo"] = lo
        if cache.root:
            cache.store(sname, lo, hi, meta["prompts"], e_img, e_txt, meta)
        return e_img, e_txt, meta

    start_t = datetime.now()
    emb = {"train": embed_split(train_set, "train")}
    # DB = train rows listed in train_indices_in_compr, in that order (Subset order, run_lemon.py:124)
    n_train = len(train_set)
    full_img = all_gather_rows(emb["train"][0], n_train)
    full_txt = all_gather_rows(emb["train"][1], n_train)
    tr_prompt_ids = ids_of(emb["train"][2])
    full_pid = all_gather_rows(torch.from_numpy(tr_prompt_ids).to(device), n_train) if world > 1 else \
        torch.from_numpy(tr_prompt_ids).to(device)
    sel = torch.from_numpy(np.asarray(train_indices_in_compr)).to(device)
    algo = {"auto": None, "f32": _lib.ALGO_F32_MFMA, "bf16": _lib.ALGO_BF16_FILTER}[args.algo]
    db = LemonDB(full_img[sel], full_txt[sel], args.dist_type, tr_label_id=full_pid[sel], algo=algo)

    cls_txt = None
    if is_clf:   # class-prompt embeddings, only used by --normalize_d1 (:180-190)
        cls_txt = embedder.embed_texts(tokenize([prompt_fn(t) for t in label_set]))

    in_db_mask = np.zeros(n_train, dtype=np.uint8)
    in_db_mask[train_indices_in_compr] = 1
    names = ["val", "test"] if (args.debug or args.skip_train) else ["train", "val", "test"]
    sets = {"train": train_set, "val": val_set, "test": test_set}
    k = args.knn_k
    frames = []
    for sname in names:
        if sname not in emb:
            emb[sname] = embed_split(sets[sname], sname)
        e_img, e_txt, meta = emb[sname]
        nq, lo = e_img.shape[0], meta["lo"]
        rec = db.neighbors(e_img, e_txt, k, drop_self=(sname == "train"),
                           in_db=in_db_mask[lo:lo + nq] if sname == "train" else None,
                           discrete=args.use_discrete_for_text, q_label_id=ids_of(meta),
                           return_indices=False)
        if args.normalize_d1:
            assert is_clf
            rec["d_1"] = ops.d1_normalized(args.dist_type, e_img, cls_txt,
                                           torch.from_numpy(np.asarray(meta["noisy"], dtype=np.int32)))
        n_total = len(sets[sname])
        host = {key: all_gather_rows(v, n_total).cpu().numpy() for key, v in rec.items()}
        flips = 1 - (np.array(meta["noisy_txt"]) == np.array(meta["clean_txt"]))
        if world > 1:   # metadata to rank 0 (small python objects)
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, (meta, flips))
            meta = {key: sum((g[0][key] for g in gathered), []) for key in ("noisy", "clean", "noisy_txt", "clean_txt")}
            flips = np.concatenate([g[1] for g in gathered])
        if rank == 0:
            frames.append(pd.DataFrame({
                "sset": sname, "idx": np.arange(n_total),
                "actual_label": [c.item() if hasattr(c, "item") else c for c in meta["clean"]],
                "actual_label_text": meta["clean_txt"], "noisy_label": list(meta["noisy"]),
                "noisy_label_text": meta["noisy_txt"], "is_mislabel": flips, "is_correct_label": 1 - flips,
                "d_1": host["d_1"].astype(np.float64),
                **{c: list(host[c]) for c in ("dists_n", "D_n", "dists_tr_n", "dists_m", "D_m", "dists_tr_m")},
            }))
    torch.cuda.synchronize(device)
    timedelta = (datetime.now() - start_t).total_seconds()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return 0
    df = pd.concat(frames, ignore_index=True)
    n_samples = len(df)
    print(f"Finished {n_samples} samples in {timedelta} seconds; avg of {timedelta / n_samples}s per sample")

    if "d1" in args.ablation:
        df["d_1"] = 0.0

    def device_rec(frame):
        return {c: torch.from_numpy(np.stack(frame[c].values).astype(np.float32)).to(device)
                for c in ("D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")} | \
               {"d_1": torch.from_numpy(frame["d_1"].values.astype(np.float32)).to(device)}

    if args.real_dataset or args.skip_hparam_optim:
        res = {"df": df}
    else:
        df_val = df.query('sset == "val"')
        rec_val = device_rec(df_val)
        y_val = df_val["is_mislabel"].values
        score_fn = lambda hp: ops.lemon_score(rec_val, hp).cpu().numpy()      # K5 on the device-resident val arrays
        grid = {"beta": np.arange(0, 100.01, 5), "gamma": np.arange(0, 100.01, 5), "tau_1": [0, 1, 5, 10],
                "tau_2": [0, 1, 5, 10]} if args.hparam_grid == "full" else \
               {"beta": [0, 5, 10], "gamma": [0, 5, 10], "tau_1": [0, 1], "tau_2": [0, 5]}
        crit = "know_val_labels"
        zero6 = dict(beta=0, gamma=0, tau_1_n=0, tau_2_n=0, tau_1_m=0, tau_2_m=0)
        if args.ablation == "only_beta":
            sel_res = {**zero6, "beta": 1}
        elif args.ablation == "only_gamma":
            sel_res = {**zero6, "gamma": 1}
        else:
            if args.ablation == "multimodal_baseline":
                best = [0] * 6
                best_f1, best_thres = M.optimize_f1_efficient(y_val, df_val["d_1"].values, return_thres=True)
            else:
                force_zero = {"none": [], "d1": [], "tau_1": ["tau_1_n", "tau_1_m"], "tau_2": ["tau_2_n", "tau_2_m"],
                              "tau_1_2": ["tau_1_n", "tau_1_m", "tau_2_n", "tau_2_m"], "beta": ["beta"],
                              "gamma": ["gamma"]}[args.ablation]
                force_one = ["beta"] if args.ablation == "d1" else []
                rec_lbfgs = {c: (df_val[c].values if c == "d_1" else np.stack(df_val[c].values))
                             for c in ("d_1", "D_n", "dists_tr_n", "dists_n", "D_m", "dists_tr_m", "dists_m")}
                best, best_f1, best_thres = M.maximize_metric(
                    score_fn, y_val, grid, [[0] * 6, [0.5] * 6, [1] * 6, [10] * 6], M.optimize_f1_efficient, {},
                    force_zero=force_zero, force_one=force_one, rec_for_lbfgs=rec_lbfgs,
                    lbfgs_device=str(device) if args.lbfgs_device == "cuda" else "cpu",
                    batch_grid=lambda hps: ops.grid_f1(rec_val, y_val, [[hp[n] for n in M.HP_NAMES] for hp in hps])[0])
            sel_res = dict(zip(M.HP_NAMES, best))
            sel_res.update(thres=best_thres, selected_val=best_f1)
        s, dn, dm = ops.lemon_score(device_rec(df), sel_res, return_dn=True)
        df[f"{crit}_pred_score"] = s.cpu().numpy()
        df[f"{crit}_d_n"], df[f"{crit}_d_m"] = dn.cpu().numpy(), dm.cpu().numpy()
        df_val = df.query('sset == "val"')
        prev = df.loc[df.sset == "val", "is_mislabel"].sum() / (df.sset == "val").sum()
        thress = M.eval_metrics(df_val["is_mislabel"], df_val[f"{crit}_pred_score"], prevalence=prev)
        for sset in df.sset.unique():
            sub = df.loc[df.sset == sset]
            sel_res[sset] = M.eval_metrics(sub["is_mislabel"], sub[f"{crit}_pred_score"], prevalence=prev,
                                           fix_thress=thress)
        df[["sset", "idx", "actual_label", "noisy_label", "is_mislabel", f"{crit}_pred_score"]].rename(
            columns={f"{crit}_pred_score": "pred_score"}).to_csv(out_dir / f"{crit}_scores.csv")
        res = {"df": df, "agg_results": {crit: sel_res}}
        for sset in df.sset.unique():
            print(f"{sset}: AUROC {sel_res[sset]['AUROC']:.4f}  F1 {sel_res[sset]['F1_optimal']:.4f}")

    pickle.dump(res, (out_dir / "res.pkl").open("wb"))
    if args.skip_hparam_optim:
        with open(os.path.join(out_dir, "need_hparam_optim"), "w") as f:
            f.write("need_hparam_optim")
    with open(os.path.join(out_dir, "done"), "w") as f:
        f.write("done")
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
