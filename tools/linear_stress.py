"""Tune lemon_linear_f32 over many small/odd shapes and report solutions the validation rejected (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LEMON_LINEAR_VERBOSE"] = "1"; os.environ["LEMON_LINEAR_TUNED"] = ""
from lemon_amd.ops import linear
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for m in (1, 7, 63, 144, 512, 1500, 4096, 8192):
    for k, n in ((48, 144), (48, 48), (48, 96), (96, 48), (40, 120), (40, 40), (40, 80), (80, 40), (512, 1536), (768, 768)):
        x = torch.randn(m, k, device=dev, generator=g); w = torch.randn(n, k, device=dev, generator=g) / k ** 0.5
        b = torch.randn(n, device=dev, generator=g); r = torch.randn(m, n, device=dev, generator=g)
        for mode in ("bias", "res", "silu"):
            y = linear(x, w, b, residual=r if mode == "res" else None, act="silu" if mode == "silu" else None)
            ref = x.double() @ w.double().T + b.double()
            if mode == "silu": ref = ref * torch.sigmoid(ref)
            if mode == "res": ref = ref + r.double()
            err = (y.double() - ref).abs().max().item()
            if not err < 1e-3:
                bad += 1; print(f"WRONG m={m} k={k} n={n} {mode}: max err {err}", flush=True)
print("wrong results:", bad)
