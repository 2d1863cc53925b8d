"""GPU: end-to-end `python -m lemon_amd.run_lemon` on synthetic data (no files needed): outputs,
record schema (run_lemon.py:291-307) and the DB-subset / self-exclusion path."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COLS = ["sset", "idx", "actual_label", "actual_label_text", "noisy_label", "noisy_label_text", "is_mislabel",
        "is_correct_label", "d_1", "dists_n", "D_n", "dists_tr_n", "dists_m", "D_m", "dists_tr_m"]


def _run(tmp_path, *extra):
    from lemon_amd.run_lemon import main
    out = str(tmp_path / "run")
    rc = main(["--output_dir", out, "--dataset", "cifar10", "--noise_type", "asymmetric", "--data_root", "synthetic:1500",
               "--clip_path", "random:tiny", "--hparam_grid", "small", "--debug", *extra])
    assert rc == 0
    return out


def test_cli_full_run_outputs_and_schema(hip, tmp_path):
    out = _run(tmp_path, "--knn_k", "5")
    for f in ("args.json", "res.pkl", "know_val_labels_scores.csv", "done"):
        assert os.path.exists(os.path.join(out, f)), f
    res = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))
    df = res["df"]
    assert list(df.columns[:len(COLS)]) == COLS
    assert set(df.sset.unique()) == {"val", "test"} and len(df) == 300          # --debug skips train
    assert df["D_n"].iloc[0].shape == (5,) and df["D_n"].iloc[0].dtype == np.float32
    assert (df["D_n"].iloc[0] <= 0).all()                                          # cosine: D_n = -IP (A14)
    agg = res["agg_results"]["know_val_labels"]
    assert {"beta", "gamma", "thres", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m", "selected_val", "val", "test"} <= set(agg)
    assert 0.0 <= agg["val"]["AUROC"] <= 1.0 and "F1_optimal" in agg["test"]
    assert 0.3 < df.is_mislabel.mean() < 0.5


def test_cli_train_split_db_subset_and_skip_hparam(hip, tmp_path):
    from lemon_amd.run_lemon import main
    out = str(tmp_path / "run2")
    rc = main(["--output_dir", out, "--dataset", "cifar100", "--noise_type", "symmetric", "--data_root", "synthetic:1000",
               "--clip_path", "random:tiny", "--knn_k", "3", "--compr_dataset_size_limit", "300", "--skip_hparam_optim",
               "--dist_type", "euclidean", "--use_discrete_for_text"])
    assert rc == 0
    res = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))
    df = res["df"]
    assert "agg_results" not in res and os.path.exists(os.path.join(out, "need_hparam_optim"))
    assert (df.sset == "train").sum() == 800 and (df.sset == "val").sum() == 100
    tr = df[df.sset == "train"]
    assert set(np.unique(np.concatenate(list(tr["dists_n"])))) <= {0.0, 1.0}       # discrete text metric
    assert os.path.exists(os.path.join(out, "out.txt"))                             # Tee (no --debug)


def test_cli_cat_noise_on_cifar_raises(hip, tmp_path):
    from lemon_amd.run_lemon import main
    with pytest.raises(NotImplementedError):
        main(["--output_dir", str(tmp_path / "x"), "--dataset", "cifar100", "--noise_type", "cat",
              "--data_root", "synthetic:200", "--clip_path", "random:tiny", "--debug"])


@pytest.mark.parametrize("discrete", [False, True])
def test_cli_synthetic_mscoco_cat_noise_matches_oracle(hip, oracle, tmp_path, discrete):
    """BASELINE configs[2] plumbing offline: mscoco surface, noise_type cat, caption-side text kNN, DB = random subset
    smaller than train (mixed in_db), tiny random CLIP.  The per-sample records of res.pkl must equal the oracle's on the
    embeddings the run itself cached (--embedding_cache), incl. the reference's DB-subset draw."""
    import glob
    import pickle
    from lemon_amd.run_lemon import main
    out, cache = str(tmp_path / "run"), str(tmp_path / "cache")
    argv = ["--output_dir", out, "--dataset", "mscoco", "--noise_type", "cat", "--noise_level", "0.4", "--data_root",
            "synthetic:3000", "--clip_path", "random:tiny", "--knn_k", "5", "--compr_dataset_size_limit", "800",
            "--skip_hparam_optim", "--embedding_cache", cache, "--seed", "4"] + (["--use_discrete_for_text"] if discrete else [])
    assert main(argv) == 0
    df = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))["df"]
    emb = {}
    for d_ in glob.glob(os.path.join(cache, "*")):
        meta = pickle.load(open(os.path.join(d_, "meta.pkl"), "rb"))
        emb[len(meta["prompts"])] = (np.load(os.path.join(d_, "img.npy")), np.load(os.path.join(d_, "txt.npy")), meta)
    n_tr = (df.sset == "train").sum()
    assert n_tr > 1500 and 0.3 < df[df.sset == "train"].is_mislabel.mean() < 0.45
    np.random.seed(4)
    sel = np.random.choice(np.arange(n_tr), 800, replace=False)                # run_lemon.py:81,123
    img_tr, txt_tr, meta_tr = emb[n_tr]
    vocab = {}
    ids = lambda prompts: np.array([vocab.setdefault(p, len(vocab)) for p in prompts], np.int32)
    tr_ids = ids(meta_tr["prompts"])
    for s in ("train", "val", "test"):
        sub = df[df.sset == s]
        q_img, q_txt, meta = emb[len(sub)]
        in_db = np.isin(np.arange(len(sub)), sel).astype(np.uint8) if s == "train" else None
        ref = oracle.neighbors("cosine", img_tr[sel], txt_tr[sel], q_img, q_txt, 5, drop_self=(s == "train"), in_db=in_db,
                               discrete=discrete, tr_label_id=tr_ids[sel], q_label_id=ids(meta["prompts"]))
        for col in ("D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
            assert np.array_equal(np.stack(sub[col].values), ref[col]), (s, col)
        assert np.array_equal(sub["d_1"].values.astype(np.float32), ref["d_1"])
    # a second run re-uses the cache (no encoder) and reproduces the records bit for bit
    out2 = str(tmp_path / "run2")
    assert main([a if a != out else out2 for a in argv]) == 0
    df2 = pickle.load(open(os.path.join(out2, "res.pkl"), "rb"))["df"]
    assert all(np.array_equal(np.stack(df[c].values), np.stack(df2[c].values)) for c in ("D_n", "dists_m", "dists_tr_n"))


def test_cli_config0_shape_vit_b32_full_protocol_matches_oracle(hip, oracle, tmp_path):
    """BASELINE configs[0]/[1] plumbing at the REAL encoder size: CIFAR-10 surface, pair-flip noise 0.4, ViT-B/32 (random
    weights: no checkpoint offline), 3 000 synthetic images through the GPU preprocessing, k = 50, the full run incl. the
    hyper-parameter search (small grid) and the output files; records == the oracle's on the cached embeddings."""
    import glob
    import pickle
    from lemon_amd.run_lemon import main
    out, cache = str(tmp_path / "run"), str(tmp_path / "cache")
    assert main(["--output_dir", out, "--dataset", "cifar10", "--noise_type", "asymmetric", "--noise_level", "0.4", "--data_root",
                 "synthetic:3000", "--clip_path", "random:vit-b-32", "--knn_k", "50", "--hparam_grid", "small", "--embedding_cache",
                 cache, "--encoder_batch", "500"]) == 0
    for f in ("args.json", "res.pkl", "know_val_labels_scores.csv", "done", "out.txt", "err.txt"):
        assert os.path.exists(os.path.join(out, f)), f
    res = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))
    df = res["df"]
    assert (df.sset == "train").sum() == 2400 and set(df.sset.unique()) == {"train", "val", "test"}
    emb = {}
    for d_ in glob.glob(os.path.join(cache, "*")):
        meta = pickle.load(open(os.path.join(d_, "meta.pkl"), "rb"))
        emb[len(meta["prompts"])] = (np.load(os.path.join(d_, "img.npy")), np.load(os.path.join(d_, "txt.npy")), meta)
    img_tr, txt_tr, _ = emb[2400]
    assert img_tr.shape == (2400, 512) and np.abs(np.linalg.norm(img_tr, axis=1) - 1).max() < 1e-5
    sub = df[df.sset == "train"]
    ref = oracle.neighbors("cosine", img_tr, txt_tr, img_tr, txt_tr, 50, drop_self=True)
    for col in ("D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
        assert np.array_equal(np.stack(sub[col].values), ref[col]), col
    agg = res["agg_results"]["know_val_labels"]
    score = oracle.score({**ref, "d_1": ref["d_1"]}, {h: agg[h] for h in ("beta", "gamma", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m")})
    assert np.abs(sub["know_val_labels_pred_score"].values - score).max() < 1e-6
    assert abs(oracle.auroc(sub["is_mislabel"].values, score) - agg["train"]["AUROC"]) < 1e-9
