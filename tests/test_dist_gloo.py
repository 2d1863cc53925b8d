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


# ---- bench.py's own launcher: the rendezvous port travels to rank 0 as an OPEN listening socket (no bind-close-reuse) ----
_HANDOFF_RANK = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
store = bench.rendezvous_store(rank, world)
assert store is not None
dist.init_process_group("gloo", rank=rank, world_size=world, store=store)
from lemon_amd.pipeline import all_gather_rows, GatherLog
t = torch.arange(3 * 4, dtype=torch.float32).reshape(3, 4) + 100 * rank
g = all_gather_rows(t, 3 * world)
assert g.shape == (3 * world, 4) and all(torch.equal(g[3 * r:3 * r + 3], t - 100 * rank + 100 * r) for r in range(world))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_launcher_hands_rank0_an_open_listening_socket():
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    lsock, port = bench.open_rendezvous_socket()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LEMON_MASTER_HANDOFF="1")
        if r == 0:
            env["LEMON_MASTER_LISTEN_FD"] = str(lsock.fileno())
        procs.append(subprocess.Popen([sys.executable, "-c", _HANDOFF_RANK.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      pass_fds=(lsock.fileno(),) if r == 0 else ()))
    lsock.close()
    outs = [p.communicate(timeout=240) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, e.decode()[-2000:])
        assert f"rank {r} ok" in o.decode()
    assert b"hand-off failed" not in outs[0][1], outs[0][1].decode()[-500:]      # the fd path itself, not the fallback


def test_one_rank_group_all_gather_is_forced_through_the_collective(tmp_path):
    """world_size 1: all_gather_rows returns the shard itself unless forced -- then it goes through all_gather_into_tensor
    (the switch bench.py --rccl_world1 and tests/test_gpu_rccl.py use to execute RCCL on a one-GPU box)."""
    import subprocess
    code = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
dist.init_process_group("gloo", rank=0, world_size=1, store=bench.rendezvous_store(0, 1))
from lemon_amd import pipeline
calls = []
orig = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
t = torch.randn(5, 3)
assert pipeline.all_gather_rows(t, 5) is t and not calls
g = pipeline.all_gather_rows(t, 5, force=True)
assert torch.equal(g, t) and g is not t and len(calls) == 1
os.environ["LEMON_FORCE_ALLGATHER"] = "1"
assert torch.equal(pipeline.all_gather_rows(t, 5), t) and len(calls) == 2
dist.destroy_process_group()
print("ok")
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "MASTER_ADDR", "RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=240)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]
