"""The two places the reference CONSUMES LEMoN scores to clean a training set: both are a selection on the
score column, restated here so the score files this package writes plug into them unchanged.

  select_cleanest          train_clip_from_scratch.py:95-114   (--cc3m_filtering DIR --cc3m_filtering_n N)
  percentile_filter_indices lib/downstream/downstream_captioning.py:229-236 (--filter_data --subset_csv_path ...)
"""
import json
import os

import numpy as np
import pandas as pd

FIXED_HPARAMS = dict(beta=5, gamma=5, tau_1_n=0.1, tau_2_n=5, tau_1_m=0.1, tau_2_m=5)   # train_clip_from_scratch.py:102-109


def select_cleanest(result_dir, n, score_fn=None):
    """`idx` values of the n samples LEAST likely to be mislabelled in a run_lemon output directory (res.pkl + args.json):
    score = d_1 for an `--ablation multimodal_baseline` run, else the LEMoN score at the fixed hyper-parameters; ascending
    sort (higher score = more likely mislabelled), first n.  score_fn(df, hparams) defaults to the HIP score kernel."""
    assert n > 0
    df = pd.read_pickle(os.path.join(result_dir, "res.pkl"))["df"].copy()
    with open(os.path.join(result_dir, "args.json")) as f:
        run_args = json.load(f)
    if run_args["ablation"] == "multimodal_baseline":
        df["score"] = df["d_1"]
    else:
        if score_fn is None:
            from .ops import calc_scores_given_hparams_vectorized as score_fn
        df["score"] = np.asarray(score_fn(df, FIXED_HPARAMS))
    return df.sort_values(by="score", ascending=True).iloc[:n]["idx"].values


def percentile_filter_indices(csv_path, percentile, split_col="sset", score_col="pred_score", split="train"):
    """Positions (within the split's rows of a `*_scores.csv`) of the samples whose score lies below the given percentile
    of that split's scores: the subset the downstream captioning fine-tune keeps.  Also returns the kept fraction."""
    df = pd.read_csv(csv_path)
    df = df[df[split_col] == split]
    cut = np.percentile(df[score_col].values, percentile)
    keep = np.asarray(df[score_col].values < cut, dtype=np.int8)
    idx = np.arange(df.shape[0])[keep == 1]
    return idx, len(idx) / max(df.shape[0], 1)
