"""What the library's fp16 GEMMs (lemon_linear_f16x3 operands, fp32 out) reach per shape, every supported solution timed
(LEMON_LINEAR_TUNE=1): is a tower shape slow because of its n / m, and would padding or another micro-batch size help?
python tools/gemm_shape_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LEMON_LINEAR_TUNED"] = ""
os.environ["LEMON_LINEAR_TUNE"] = "1"
os.environ["LEMON_LINEAR_ALLOW_POSITION_DEPENDENT"] = "1"
os.environ.setdefault("LEMON_LINEAR_TUNE_MS", "4000")
import torch
from lemon_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def best(m, n, k3, bias=True):
    x = (torch.randn((m, k3), device=dev, generator=g)).half()
    w = (torch.randn((n, k3), device=dev, generator=g) * 100).half()
    b = torch.randn(n, device=dev, generator=g) if bias else None
    ops.linear_split(x, w, b)           # tunes the key
    torch.cuda.synchronize()
    t = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.linear_split(x, w, b)
        e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / 5)
    return min(t)


shapes = [(50000, n, 2304) for n in (768, 1536, 2048, 2304, 2560, 3072, 4608)] + \
         [(m, 2304, 2304) for m in (25000, 49152, 50176, 51200, 100000)] + \
         [(50000, 768, 9216), (50000, 1536, 9216), (100000, 768, 2304), (100000, 768, 9216)]
for (m, n, k3) in shapes:
    ms = best(m, n, k3)
    print(f"m={m:6d} n={n:5d} k3={k3:5d}: {ms*1e3:8.1f} us  {2.0*m*n*k3/ms/1e9:7.1f} TFLOP/s fp16", flush=True)
