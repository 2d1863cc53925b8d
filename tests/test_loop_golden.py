"""CPU: the oracle's per-sample loop against fixtures produced by the REFERENCE's own loop
(tools/make_golden_loop.py: /root/reference/run_lemon.py under runpy with stand-ins for faiss / weights / data).

Pinned here: self-exclusion (run_lemon.py:257-263,277-283), the D_n / D_m sign quirk (:269-270,285-286), the discrete
text metric (:266-267), --normalize_d1 (:244-248), the DB subset + mixed in_db (:122-124,258), split order and the
record schema (:291-307), score aggregation + AUROC on the reference's own frames.  NOT pinned: faiss's arithmetic
and tie order (the search stand-in implements DESIGN.md's documented chain / lowest-index rule)."""
import numpy as np
import pytest

from tests.loopfx import LoopCase, assert_records_match, case_names

CASES = case_names()


def test_fixtures_present():
    assert len(CASES) >= 12, "run tools/make_golden_loop.py in the build container"


@pytest.mark.parametrize("name", CASES)
def test_oracle_search_equals_reference_side_search(oracle, name):
    """oracle.knn (C, fmaf chain) == the numpy stand-in the reference loop ran on (emulated fma chain,
    lexicographic (score, index) order): two independent implementations of the documented contract."""
    c = LoopCase(name)
    db_img, db_txt = c.db()
    for s in c.ssets:
        q_img, q_txt = c.queries(s)
        ks = c.k + (s == "train")
        for X, Q, side in ((db_img, q_img, "img"), (db_txt, q_txt, "txt")):
            D, I = oracle.knn(c.metric, X, Q, ks)
            assert np.array_equal(D, c.fx[f"{s}_search_D_{side}"]), (name, s, side)
            assert np.array_equal(I, c.fx[f"{s}_search_I_{side}"]), (name, s, side)


@pytest.mark.parametrize("name", CASES)
def test_oracle_neighbors_equals_reference_loop(oracle, name):
    c = LoopCase(name)
    db_img, db_txt = c.db()
    for s in c.ssets:
        q_img, q_txt = c.queries(s)
        tr_lab, q_lab = c.label_ids(s) if c.discrete else (None, None)
        out = oracle.neighbors(c.metric, db_img, db_txt, q_img, q_txt, c.k, drop_self=(s == "train"),
                               in_db=c.in_db(s), discrete=c.discrete, tr_label_id=tr_lab, q_label_id=q_lab)
        if c.normalize_d1:
            out["d_1"] = oracle.d1_normalized(c.metric, q_img, c.fx["cls_txt"], c.fx[f"{s}_noisy"])
        assert_records_match(out, c, s)
        assert np.abs(out["dists_tr"] - c.fx["dists_tr"]).max() <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_db_is_the_normalised_train_subset(oracle, name):
    """DB rows = normalize(train embeddings)[train_indices_in_compr] in that order (run_lemon.py:122-127,163-164);
    queries of the train split = the same normalised rows (the reference embeds train twice, same values)."""
    c = LoopCase(name)
    db_img, db_txt = c.db()
    assert len(db_img) == len(c.sel) == min(c.n_train, len(db_img))
    if "train" in c.ssets:
        q_img, q_txt = c.queries("train")
        assert np.array_equal(q_img[c.sel], db_img) and np.array_equal(q_txt[c.sel], db_txt)
    nrm = np.abs(np.linalg.norm(db_img.astype(np.float64), axis=1) - 1).max()
    assert nrm < 1e-6
    # our normalisation of the same raw rows (oracle, float64 accumulation) vs the reference's F.normalize
    raw = c.fx["img_all"]
    ours = oracle.normalize_rows(raw)
    # every DB row must be one of the normalised raw rows (order is the split's business, checked by the CLI test)
    gap = np.abs(db_img[:50, None, :] - ours[None, :, :]).max(-1).min(-1)
    assert gap.max() <= 2e-7, gap.max()


@pytest.mark.parametrize("name", [n for n in CASES if LoopCase(n).agg is not None])
def test_scores_and_auroc_on_reference_frames(oracle, name):
    """oracle.score / oracle.auroc on the reference's own per-sample arrays == the reference's pred_score column
    and agg_results AUROC (lib/metrics/utils.py:47-82,408-412 executed by the reference run itself)."""
    c = LoopCase(name)
    hp = {k_: c.agg[k_] for k_ in ("beta", "gamma", "tau_1_n", "tau_2_n", "tau_1_m", "tau_2_m")}
    lo = 0
    for s in c.ssets:
        rec = c.expected(s)
        n = len(rec["d_1"])
        sc, dn, dm = oracle.score(rec, hp, return_dn=True)
        sc = sc - rec["d_1"].astype(np.float32).astype(np.float64) + rec["d_1"]     # d_1 is float64 in the frame
        ref = c.fx["pred_score"][lo:lo + n]
        assert np.allclose(sc, ref, rtol=1e-6, atol=1e-9), (name, s, np.abs(sc - ref).max())
        assert np.allclose(dn, c.fx["pred_d_n"][lo:lo + n], rtol=1e-6, atol=1e-9)
        assert np.allclose(dm, c.fx["pred_d_m"][lo:lo + n], rtol=1e-6, atol=1e-9)
        assert abs(oracle.auroc(c.fx[f"{s}_is_mislabel"], ref) - c.agg[s]["AUROC"]) < 1e-12
        lo += n
