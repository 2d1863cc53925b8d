"""world_size-2 gloo tests (CPU) of the N>1 host path: shard bounds, the all-gather of ragged DB
shards in global row order, and that query-sharded scoring over the gathered DB equals the
single-process result (checked with the CPU oracle standing in for the device kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tr, n_q, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lemon_amd.pipeline import all_gather_rows, shard_bounds
        from oracle import oracle as o
        from tests.synth import planted
        s = planted(seed=5, n_tr=n_tr, n_q=n_q, d=32, C=8)
        img_tr, txt_tr, _, _ = s["train"]
        q_img, q_txt, _, _ = s["query"]
        lo, hi = shard_bounds(n_tr, world, rank)
        g_img = all_gather_rows(torch.from_numpy(img_tr[lo:hi]), n_tr).numpy()
        g_txt = all_gather_rows(torch.from_numpy(txt_tr[lo:hi]), n_tr).numpy()
        g_lab = all_gather_rows(torch.arange(lo, hi, dtype=torch.int32), n_tr).numpy()
        assert np.array_equal(g_img, img_tr) and np.array_equal(g_txt, txt_tr)
        assert np.array_equal(g_lab, np.arange(n_tr, dtype=np.int32))          # global row order
        qlo, qhi = shard_bounds(n_q, world, rank)
        rec = o.neighbors("cosine", g_img, g_txt, q_img[qlo:qhi], q_txt[qlo:qhi], 5)
        np.savez(os.path.join(tmp, f"r{rank}.npz"), lo=qlo, hi=qhi, **{k: v for k, v in rec.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tr,n_q", [(301, 77), (64, 2), (5, 1)])
def test_sharded_pipeline_matches_single_process(tmp_path, n_tr, n_q):
    from oracle import oracle as o
    from tests.synth import planted
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_tr, n_q, str(tmp_path)), nprocs=world, join=True)
    s = planted(seed=5, n_tr=n_tr, n_q=n_q, d=32, C=8)
    img_tr, txt_tr, _, _ = s["train"]
    q_img, q_txt, _, _ = s["query"]
    ref = o.neighbors("cosine", img_tr, txt_tr, q_img, q_txt, 5)
    rows = 0
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        rows += hi - lo
        for key in ("I_n", "I_m", "d_1", "D_n", "dists_n", "dists_tr_n", "D_m", "dists_m", "dists_tr_m"):
            assert np.array_equal(z[key], ref[key][lo:hi], equal_nan=True), (r, key)
    assert rows == n_q


def test_shard_bounds_cover_everything():
    from lemon_amd.pipeline import shard_bounds
    for n in (0, 1, 7, 8, 9, 40000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            per = (n + w - 1) // w
            assert all(hi - lo <= per for lo, hi in spans)
