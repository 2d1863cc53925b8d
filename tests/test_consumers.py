"""CPU: the score consumers (train_clip_from_scratch.py:95-114, lib/downstream/downstream_captioning.py:229-236) on a frame
taken from a reference-run fixture, against the literal pandas statements of those call sites."""
import json
import pickle

import numpy as np
import pandas as pd

from lemon_amd import consumers
from tests.loopfx import REC, LoopCase


def _frame(case):
    rows = []
    for s in case.ssets:
        exp = case.expected(s)
        n = len(exp["d_1"])
        d = {"sset": s, "idx": np.arange(n), "d_1": exp["d_1"], "is_mislabel": case.fx[f"{s}_is_mislabel"]}
        for c in REC:
            d[c] = list(exp[c])
        rows.append(pd.DataFrame(d))
    return pd.concat(rows, ignore_index=True)


def test_select_cleanest_and_percentile_filter(oracle, tmp_path):
    case = LoopCase("c10_cos_k5_full")
    df = _frame(case)
    for ablation in ("none", "multimodal_baseline"):
        out = tmp_path / ablation
        out.mkdir()
        pickle.dump({"df": df}, open(out / "res.pkl", "wb"))
        json.dump({"ablation": ablation}, open(out / "args.json", "w"))
        score_fn = lambda frame, hp: oracle.score({c: np.stack(frame[c].values) if c != "d_1" else frame[c].values.astype(np.float32)
                                                   for c in REC + ("d_1",)}, hp)
        got = consumers.select_cleanest(str(out), 50, score_fn=score_fn)
        ref = df.copy()
        ref["score"] = ref["d_1"] if ablation == "multimodal_baseline" else score_fn(ref, consumers.FIXED_HPARAMS)
        exp = ref.sort_values(by="score", ascending=True).iloc[:50]["idx"].values          # train_clip_from_scratch.py:112
        assert np.array_equal(got, exp)
    csv = tmp_path / "scores.csv"
    df.assign(pred_score=case.fx["pred_score"])[["sset", "idx", "is_mislabel", "pred_score"]].to_csv(csv)
    idx, frac = consumers.percentile_filter_indices(str(csv), 60)
    tr = pd.read_csv(csv)
    tr = tr[tr["sset"] == "train"]
    keep = np.array(tr["pred_score"].values < np.percentile(tr["pred_score"].values, 60), dtype=np.int8)   # :231-234
    assert np.array_equal(idx, np.arange(tr.shape[0])[keep == 1]) and abs(frac - 0.6) < 0.05
    # the kept subset is cleaner than the split
    assert tr["is_mislabel"].values[idx].mean() < tr["is_mislabel"].mean()
