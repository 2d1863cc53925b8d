"""Go/no-go probe for a 3-way split-bf16 GEMM (x = hi + mid + lo in bf16; the six products with i + j <= 2 as ONE bf16 GEMM over
a 6K-long concatenated k axis, fp32 accumulate) against the fp32 GEMM the towers run today: sustained rate of the library
bf16 GEMM on RANDOM operands at the four ViT-B/32 tower shapes, the cost of the split pass, and the error of the emulation
against float64.  (Experiment: VERDICT r2 item 8.)   python tools/split_gemm_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def split3(x):
    hi = x.to(torch.bfloat16)
    r = x - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def cat_a(x):       # [M, 6K]: hi hi mid hi mid lo
    hi, mid, lo = split3(x)
    return torch.cat([hi, hi, mid, hi, mid, lo], dim=1).contiguous()


def cat_b(w):       # w [N, K] -> [N, 6K]: hi mid hi lo mid hi   (pairs: hh, hm, mh, hl, mm, lh)
    hi, mid, lo = split3(w)
    return torch.cat([hi, mid, hi, lo, mid, hi], dim=1).contiguous()


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print(f"{'shape (m,n,k)':26s} {'fp32 ms':>8s} {'TF':>6s} | {'bf16 6K ms':>10s} {'TF-eq':>6s} {'bf16 TF':>8s} | {'split ms':>8s} | speed-up (incl. split) | max rel err emu / fp32")
for (m, n, k) in [(50000, 2304, 768), (50000, 768, 768), (50000, 3072, 768), (50000, 768, 3072), (32000, 1536, 512), (32000, 2048, 512)]:
    x = torch.randn((m, k), device=dev, generator=g)
    w = torch.randn((n, k), device=dev, generator=g) * k ** -0.5
    t32 = timeit(lambda: torch.nn.functional.linear(x, w))
    a6, b6 = cat_a(x), cat_b(w)
    t16 = timeit(lambda: torch.nn.functional.linear(a6, b6))
    tsp = timeit(lambda: cat_a(x), 3)
    flop = 2.0 * m * n * k
    # accuracy on a slice, against float64
    xs, ws = x[:512], w[:256]
    ref = xs.double() @ ws.double().T
    emu = (cat_a(xs).float().double() @ cat_b(ws).float().double().T)       # exact products of the split operands
    y32 = torch.nn.functional.linear(xs, ws).double()
    y16 = torch.nn.functional.linear(cat_a(xs), cat_b(ws)).double()         # library result (bf16 output: rounding dominates)
    scale = ref.abs().max()
    print(f"{str((m, n, k)):26s} {t32*1e3:8.3f} {flop/t32/1e12:6.1f} | {t16*1e3:10.3f} {flop/t16/1e12:6.1f} {6*flop/t16/1e12:8.1f} | {tsp*1e3:8.3f} | "
          f"{t32/t16:5.2f}x ({t32/(t16+tsp):4.2f}x) | emu-exact {float((emu-ref).abs().max()/scale):.2e}  fp32 {float((y32-ref).abs().max()/scale):.2e}", flush=True)


# ---- 2-way fp16 split, three products (lemon_linear_f16x3), through the library (first-ranked hipBLASLt solution unless tuned) ----
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd import ops
print()
print(f"{'shape (m,n,k)':26s} {'fp32 lib ms':>11s} | {'bf16x6 ms':>9s} {'f16x3 ms':>9s} | {'split6 ms':>9s} {'split3 ms':>9s} | max rel err vs float64: f16x3 / bf16x6 / fp32 GEMM")
for (m, n, k) in [(50000, 2304, 768), (50000, 768, 768), (50000, 3072, 768), (50000, 768, 3072), (32000, 1536, 512), (32000, 2048, 512)]:
    x = torch.randn((m, k), device=dev, generator=g)
    w = torch.randn((n, k), device=dev, generator=g) * k ** -0.5
    ws = ops.weight_scale_f16x3(w)
    x6, w6 = ops.split_operand(x, "bf16x6"), ops.split_operand(w, "bf16x6", weight=True)
    x3, w3 = ops.split_operand(x, "f16x3"), ops.split_operand(w, "f16x3", weight=True, wscale=ws)
    t32 = timeit(lambda: ops.linear(x, w))
    t6 = timeit(lambda: ops.linear_split(x6, w6))
    t3 = timeit(lambda: ops.linear_split(x3, w3, alpha=1.0 / ws))
    ts6 = timeit(lambda: ops.split_operand(x, "bf16x6"), 3)
    ts3 = timeit(lambda: ops.split_operand(x, "f16x3"), 3)
    ref = x[:2048].double() @ w.double().T
    scale = ref.abs().max()
    e3 = float((ops.linear_split(x3[:2048], w3, alpha=1.0 / ws).double() - ref).abs().max() / scale)
    e6 = float((ops.linear_split(x6[:2048], w6).double() - ref).abs().max() / scale)
    e32 = float((ops.linear(x[:2048], w).double() - ref).abs().max() / scale)
    r3 = float((ops.linear_split(x3[:2048], w3, alpha=1.0 / ws).double() - ref).pow(2).mean().sqrt() / scale)
    r32 = float((ops.linear(x[:2048], w).double() - ref).pow(2).mean().sqrt() / scale)
    print(f"{str((m, n, k)):26s} {t32*1e3:11.3f} | {t6*1e3:9.3f} {t3*1e3:9.3f} | {ts6*1e3:9.3f} {ts3*1e3:9.3f} | {e3:.2e} / {e6:.2e} / {e32:.2e}   rms f16x3 {r3:.2e} fp32 {r32:.2e}", flush=True)
