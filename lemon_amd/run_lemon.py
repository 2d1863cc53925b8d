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
import os
import pickle
import sys
from datetime import datetime

import numpy as np
import pandas as pd
import torch

from .cli_common import CLF_DATASETS, add_extension_flags  # noqa: F401  (CLF_DATASETS re-exported: run_lemon.py:32)


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
    add_extension_flags(p)
    p.add_argument("--lbfgs_device", default="cuda", choices=["cuda", "cpu"],
                   help="where the SoftMargin/LBFGS polish of the hyper-parameter search runs (lib/metrics/utils.py:121-149)")
    p.add_argument("--hparam_grid", default="full", choices=["full", "small"],
                   help="'small' = 3x3x2x2 grid for smoke runs (reference grid is 21x21x4x4)")
    return p


def main(argv=None):
    from .cli_common import run_with_tee
    return run_with_tee(_run, build_parser().parse_args(argv))


def _run(args):
    from . import _lib, metrics as M, ops
    from .cli_common import meta_columns, prepare
    from .neighbors import LemonDB
    from .pipeline import all_gather_rows

    if args.real_dataset:
        assert args.noise_level == 0.0
    ctx = prepare(args)
    out_dir, world, rank, device = ctx.out_dir, ctx.world, ctx.rank, ctx.device
    is_clf, label_set, embedder = ctx.is_clf, ctx.label_set, ctx.embedder
    embed_split, ids_of, sets = ctx.embed_split, ctx.ids_of, ctx.sets
    train_set, train_indices_in_compr = sets["train"], ctx.train_indices_in_compr

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
        cls_txt = embedder.embed_texts(ctx.tokenize([ctx.prompt_fn(t) for t in label_set]))

    in_db_mask = np.zeros(n_train, dtype=np.uint8)
    in_db_mask[train_indices_in_compr] = 1
    names = ["val", "test"] if (args.debug or args.skip_train) else ["train", "val", "test"]
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
        meta, flips = ctx.gather_meta(meta)       # metadata to every rank (small python objects)
        if rank == 0:
            frames.append(pd.DataFrame({
                **meta_columns(sname, n_total, meta, flips),
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
