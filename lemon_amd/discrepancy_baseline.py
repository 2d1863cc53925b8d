"""python -m lemon_amd.discrepancy_baseline -- drop-in for the reference's
`python -m lib.baselines.discrepancy_baseline` (lib/baselines/discrepancy_baseline.py), the baseline from
"Emphasizing Complementary Samples for Non-literal Cross-modal Retrieval".

Same flags, defaults and choices (:33-50), same outputs (args.json, out.txt, err.txt, scores.csv unless --skip_train,
res.pkl {'df','agg_results'}, done) and the same record columns (:231-241).  The embedding pass is run_lemon's
(lemon_amd/cli_common.py); the four scores -- dis_x / dis_y second-order neighbours through the DB self-kNN cache
(:164-169,213-220), div_x / div_y k x k neighbour Gram sums (:221-226) -- are one lemon_discrepancy launch per split."""
import argparse
import os
import pickle
import sys

import numpy as np
import pandas as pd
import torch

from .cli_common import add_extension_flags


def build_parser():
    p = argparse.ArgumentParser(description="Baseline from ``Emphasizing Complementary Samples for Non-literal Cross-modal Retrieval``")
    p.add_argument("--exp_name", type=str)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--dataset", type=str, default="cifar100",
                   choices=["cifar10", "cifar100", "flickr30k", "mscoco", "mimiccxr_caption", "mmimdb", "cifar10_full",
                            "cifar100_full", "mini_imagenet", "stanford_cars", "cc3m"])
    p.add_argument("--noise_type", type=str, default="real",
                   choices=["real", "asymmetric", "symmetric", "random", "noun", "cat"])
    p.add_argument("--method", type=str, default="dis_x", choices=["dis_x", "dis_y", "div_x", "div_y"])
    p.add_argument("--noise_level", type=float, default=0.4)
    p.add_argument("--clip_model", type=str, default="huggingface_clip", choices=["huggingface_clip", "biomed_clip"])
    p.add_argument("--knn_k", default=5, type=int)
    p.add_argument("--batch_size", default=128, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--data_seed", default=0, type=int)
    p.add_argument("--compr_dataset_size_limit", default=50000, type=int)
    p.add_argument("--custom_cifar_prompt", default=None)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--skip_train", action="store_true")
    add_extension_flags(p)          # local weights / data (not in the reference)
    return p


def main(argv=None):
    from .cli_common import run_with_tee
    return run_with_tee(_run, build_parser().parse_args(argv))


def _run(args):
    from . import metrics as M
    from .baselines import discrepancy_scores
    from .cli_common import meta_columns, prepare
    from .neighbors import LemonDB
    from .pipeline import all_gather_rows

    ctx = prepare(args)
    rank, world, device, sets = ctx.rank, ctx.world, ctx.device, ctx.sets
    emb = {"train": ctx.embed_split(sets["train"], "train")}
    n_train = len(sets["train"])
    sel = torch.from_numpy(np.asarray(ctx.train_indices_in_compr)).to(device)
    db = LemonDB(all_gather_rows(emb["train"][0], n_train)[sel], all_gather_rows(emb["train"][1], n_train)[sel], "cosine")  # :150-155
    names = ["val", "test"] if (args.debug or args.skip_train) else ["train", "val", "test"]
    frames = []
    for sname in names:
        if sname not in emb:
            emb[sname] = ctx.embed_split(sets[sname], sname)
        e_img, e_txt, meta = emb[sname]
        # NOTE (:209): the train split searches k+1 and keeps ALL k+1 neighbours (no self-exclusion of the query here;
        # only the cache of second-order neighbours drops self, :167-169) -- lemon_discrepancy reproduces that
        score = discrepancy_scores(db, e_img, e_txt, args.knn_k, args.method, is_train=(sname == "train"))
        n_total = len(sets[sname])
        host = all_gather_rows(score, n_total).cpu().numpy()
        meta, flips = ctx.gather_meta(meta)
        if rank == 0:
            frames.append(pd.DataFrame({**meta_columns(sname, n_total, meta, flips), "pred_score": host.astype(np.float64)}))
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.destroy_process_group()
    if rank != 0:
        return 0
    df = pd.concat(frames, ignore_index=True)
    out_dir = ctx.out_dir
    if not args.skip_train:
        df.to_csv(out_dir / "scores.csv", index=False)
    df_val = df.query('sset == "val"')
    prev = df.loc[df.sset == "val", "is_mislabel"].sum() / (df.sset == "val").sum()
    thress = M.eval_metrics(df_val["is_mislabel"], df_val["pred_score"], prevalence=prev)
    selection_results = {}
    for sset in df.sset.unique():
        sub = df.loc[df.sset == sset]
        selection_results[sset] = M.eval_metrics(sub["is_mislabel"], sub["pred_score"], prevalence=prev, fix_thress=thress)
        print(f"{sset}: AUROC {selection_results[sset]['AUROC']:.4f}  F1 {selection_results[sset]['F1_optimal']:.4f}")
    pickle.dump({"df": df, "agg_results": selection_results}, (out_dir / "res.pkl").open("wb"))
    with open(os.path.join(out_dir, "done"), "w") as f:
        f.write("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
