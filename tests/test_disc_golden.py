"""Discrepancy baselines (lib/baselines/discrepancy_baseline.py:164-242) against fixtures produced by the REFERENCE script
itself (tools/make_golden_loop.py --disc: runpy + the same stand-ins as the loop fixtures): dis_x / dis_y second-order
neighbours through the DB self-kNN cache, div_x / div_y neighbour Gram sums, incl. the train split's k+1 search that
keeps all k+1 neighbours.  CPU: the oracle; GPU: lemon_discrepancy and the `lemon_amd.discrepancy_baseline` CLI."""
import glob
import json
import os
import pickle

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
METHODS = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(G, "disc_*.npz")))


def load(m):
    fx = np.load(os.path.join(G, f"disc_{m}.npz"))
    argv = json.loads(str(fx["argv"]))
    return fx, argv, int(argv[argv.index("--knn_k") + 1])


def test_fixtures_present():
    assert METHODS == ["dis_x", "dis_y", "div_x", "div_y"]


@pytest.mark.parametrize("m", METHODS)
def test_oracle_discrepancy_equals_reference_script(oracle, m):
    fx, argv, k = load(m)
    E = fx["db_img"] if m.endswith("_x") else fx["db_txt"]
    for s in fx["ssets"]:
        s = str(s)
        q_img = oracle.normalize_rows(fx[f"{s}_q_img_raw"])
        qv = q_img if m.endswith("_x") else fx[f"{s}_q_txt"]
        got = oracle.discrepancy(m[:3], E, fx["db_txt"], qv, fx[f"{s}_q_txt"], k, is_train=(s == "train"))
        assert np.abs(got - fx[f"{s}_pred_score"]).max() <= 2e-6, (m, s)
        auroc = json.loads(str(fx["auroc"]))[s]
        assert abs(oracle.auroc(fx[f"{s}_is_mislabel"], fx[f"{s}_pred_score"]) - auroc) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("m", METHODS)
def test_lemon_discrepancy_equals_reference_script(hip, m):
    import torch
    from lemon_amd.baselines import discrepancy_scores
    from lemon_amd.neighbors import LemonDB
    from lemon_amd.ops import normalize_vectors
    fx, argv, k = load(m)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    db = LemonDB(t(fx["db_img"]), t(fx["db_txt"]), "cosine")
    for s in fx["ssets"]:
        s = str(s)
        q_img = normalize_vectors(t(fx[f"{s}_q_img_raw"]))
        got = discrepancy_scores(db, q_img, t(fx[f"{s}_q_txt"]), k, m, is_train=(s == "train")).cpu().numpy()
        assert np.abs(got - fx[f"{s}_pred_score"]).max() <= 2e-6, (m, s)


@pytest.mark.gpu
@pytest.mark.parametrize("m", METHODS)
def test_discrepancy_cli_reproduces_reference_run(hip, m, monkeypatch, tmp_path):
    from types import SimpleNamespace
    from lemon_amd.discrepancy_baseline import main
    from tests import planted
    fx, argv, k = load(m)
    case = SimpleNamespace(fx=fx, is_caption=False, dataset="cifar10")
    extra = planted.install(case, monkeypatch, tmp_path)
    out = str(tmp_path / "out")
    assert main(["--output_dir", out] + argv + extra) == 0
    assert sorted(os.listdir(out)) == sorted(str(f) for f in fx["out_files"])
    res = pickle.load(open(os.path.join(out, "res.pkl"), "rb"))
    df = res["df"]
    assert list(df.columns) == ["sset", "idx", "actual_label", "actual_label_text", "noisy_label", "noisy_label_text",
                                "is_mislabel", "is_correct_label", "pred_score"]
    auroc = json.loads(str(fx["auroc"]))
    for s in fx["ssets"]:
        s = str(s)
        sub = df[df.sset == s]
        assert np.array_equal(sub["is_mislabel"].values.astype(np.int64), fx[f"{s}_is_mislabel"])
        assert np.array_equal(np.array([int(v) for v in sub["noisy_label"]]), fx[f"{s}_noisy"])
        assert np.abs(sub["pred_score"].values - fx[f"{s}_pred_score"]).max() <= 2e-6, (m, s)
        # our metric code on the REFERENCE's scores reproduces the reference's AUROC exactly ...
        from lemon_amd import metrics as M
        assert abs(M.prob_metrics(fx[f"{s}_is_mislabel"], fx[f"{s}_pred_score"])["AUROC"] - auroc[s]) < 1e-12
        # ... and on our own scores for the image-side methods.  The text-side methods (*_y) give every sample of a class
        # the same score up to the last bit: a handful of distinct values whose ORDER decides the AUROC, so a 1e-7
        # difference between two class scores moves it by several points in the reference as much as here.
        if m.endswith("_x"):
            assert abs(res["agg_results"][s]["AUROC"] - auroc[s]) < 5e-4
