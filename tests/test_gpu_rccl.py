"""GPU: RCCL executes (SURVEY 8e steps 2-4 on a one-GPU box).  A ONE-rank "nccl" process group -- ncclCommInitRank and
ncclAllGather are the same library calls at any world size -- with the DB all-gather forced through the collective
(pipeline.all_gather_rows(force=True)); once in a fresh child process (the group is created before anything else touches the
GPU there) and once through bench.py --rccl_world1, whose line must carry the `exchange` report.  The 1/2/4/8 curve itself
needs a node and is the driver's to run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, store=bench.rendezvous_store(0, 1))
from lemon_amd.pipeline import GatherLog, all_gather_rows
log = GatherLog()
g = torch.Generator(device=dev).manual_seed(3)
for name, shape, dtype in (("emb_img_tr", (40000, 512), torch.float32), ("label_id_tr", (40000,), torch.int32), ("ragged", (777, 768), torch.float32)):
    t = (torch.randn(shape, device=dev, generator=g) * 100).to(dtype)
    out = all_gather_rows(t, shape[0], log=log, name=name, force=True)
    torch.cuda.synchronize()
    assert out.data_ptr() != t.data_ptr() and torch.equal(out, t), name
rep = log.summary()
assert len(rep["arrays"]) == 3 and all(a["ms"] > 0 and a["backend"] == "nccl" for a in rep["arrays"]), rep
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "librccl is not mapped into the process"
dist.destroy_process_group()
print("RCCL_OK", [round(a["ms"], 3) for a in rep["arrays"]])
""" % ROOT


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "MASTER_ADDR", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_all_gather_on_one_rank(hip):
    r = subprocess.run([sys.executable, "-c", _CHILD], env=_env(), capture_output=True, timeout=600)
    assert r.returncode == 0 and b"RCCL_OK" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-3000:])


def test_bench_rccl_world1_reports_the_exchange(hip):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--rccl_world1",
                        "--n_train", "4000", "--n_val", "500", "--n_test", "500", "--encoder_batch", "500", "--knn_k", "5",
                        "--no_cpu_baseline", "--no_knn_1m", "--no_f32_gemm_check"], env=_env(), capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    ex = line["exchange"]
    assert ex["backend"] == "nccl" and ex["allgather_ms"] > 0
    assert {a["name"] for a in ex["arrays"]} >= {"emb_img_tr", "emb_txt_tr"}
    assert all(a["backend"] == "nccl" and a["ms"] > 0 for a in ex["arrays"])


def test_rccl_in_this_process(hip):
    """The same one-rank group inside the test process itself, so that librccl shows among the libraries this process loaded."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    from lemon_amd.pipeline import all_gather_rows
    assert not dist.is_initialized()
    saved = {k: os.environ.pop(k) for k in ("MASTER_PORT", "MASTER_ADDR", "RANK", "WORLD_SIZE") if k in os.environ}
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, store=bench.rendezvous_store(0, 1))
    try:
        t = torch.randn(1000, 512, device=dev)
        out = all_gather_rows(t, 1000, force=True)
        torch.cuda.synchronize()
        assert torch.equal(out, t) and out.data_ptr() != t.data_ptr()
        assert "librccl" in open("/proc/self/maps").read()
    finally:
        dist.destroy_process_group()
        os.environ.update(saved)
