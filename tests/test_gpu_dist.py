"""GPU: the sharded PRODUCT path (HIP kernels, not the oracle) with two rank processes sharing cuda:0
(LEMON_DIST_BACKEND=gloo; RCCL itself needs distinct devices, the driver's 8-GPU run covers that): run_lemon's
WORLD_SIZE>1 branch -- contiguous shards, all_gather_rows of the DB shards / label ids / per-sample records, in_db offsets,
all_gather_object of the metadata -- must reproduce the single-process run bit for bit, and both must match the
REFERENCE's own run recorded in the loop fixture.  SURVEY 8e steps 1-4."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.loopfx import REC, LoopCase

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(case, out_dir, data_dir, world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LEMON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k_)
        # every rank writes to its own FILE: with pipes read one after the other a chatty rank could fill its pipe while the
        # rank being waited on sits in a collective with it
        os.makedirs(out_dir, exist_ok=True)
        log = open(os.path.join(out_dir, f"rank{r}.log"), "wb")
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), case, out_dir, data_dir],
                                       env=env, stdout=log, stderr=subprocess.STDOUT), log))
    for pr, log in procs:
        pr.wait(timeout=600)
        log.close()
    outs = [open(os.path.join(out_dir, f"rank{r}.log")).read() for r in range(world)]
    assert all(pr.returncode == 0 for pr, _ in procs), "\n".join(outs)[-3000:]
    return pickle.load(open(os.path.join(out_dir, "res.pkl"), "rb"))["df"]


@pytest.mark.parametrize("case", ["c10_cos_k5_subset", "coco_l2_k5_random_discrete", "c10_l2_k50_subset_discrete"])
def test_two_ranks_on_one_card_equal_one_rank_and_the_reference(hip, case, tmp_path):
    c = LoopCase(case)
    df1 = _launch(case, str(tmp_path / "w1"), str(tmp_path / "d1"), 1)
    df2 = _launch(case, str(tmp_path / "w2"), str(tmp_path / "d2"), 2)
    assert len(df1) == len(df2) and list(df1.columns) == list(df2.columns)
    for col in ("sset", "idx", "is_mislabel", "noisy_label_text", "actual_label_text"):
        assert (df1[col].values == df2[col].values).all(), col
    assert np.array_equal(df1["d_1"].values, df2["d_1"].values)
    for col in REC:
        assert np.array_equal(np.stack(df1[col].values), np.stack(df2[col].values)), col   # bit-identical across shardings
    for s in c.ssets:                                                                        # and == the reference's run
        sub = df2[df2.sset == s]
        exp = c.expected(s)
        for col in REC + ("d_1",):
            got = np.stack(sub[col].values) if col != "d_1" else sub[col].values
            assert np.abs(got.astype(np.float64) - exp[col]).max() <= 2e-6, (s, col)


def test_bench_self_launches_its_ranks(hip):
    """`python bench.py --gpus 2` started by hand (no torchrun, no WORLD_SIZE): the parent spawns the two ranks itself before
    touching the GPU and relays rank 0's JSON line; here both ranks share cuda:0 over gloo."""
    import json
    env = dict(os.environ, LEMON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k_, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--arch",
                          "tiny", "--n_train", "1500", "--n_val", "200", "--n_test", "200", "--knn_k", "5", "--encoder_batch", "256",
                          "--no_cpu_baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "knn_1m" not in line and "cpu_baseline" not in line          # N = 1 extras only
    assert line["config"]["parallelism"] == "dp2+allgather"
    # N > 1: the line explains its exchange step: time and bus bandwidth per gathered array, scan per rank
    ex = line["exchange"]
    assert {a["name"] for a in ex["arrays"]} == {"emb_img_tr", "emb_txt_tr"} and ex["allgather_ms"] > 0
    assert all(a["bytes_gathered"] == 2 * 1500 * 32 * 4 and a["busbw_GBs"] > 0 for a in ex["arrays"])
    assert len(ex["knn_algo_per_rank"]) == 2 and all(r["db_rows"] == 3000 for r in ex["knn_algo_per_rank"])
    # WORLD_SIZE that contradicts --gpus is an error, not a silent single-GPU run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0
