"""lemon_linear_f32 vs torch F.linear (+ separate element-wise passes) on the ViT-B/32 shapes (tuning aid)."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd.ops import linear, quick_gelu_, linear_dump_tuned
dev = torch.device("cuda:0"); M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
def bench(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, K, N, mode in [("qkv", 768, 2304, "bias"), ("out", 768, 768, "res"), ("fc1", 768, 3072, "gelu"), ("fc2", 3072, 768, "res")]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev)
    if mode == "bias":
        ref = lambda: F.linear(x, w, b); new = lambda: linear(x, w, b)
    elif mode == "res":
        ref = lambda: r + F.linear(x, w, b); new = lambda: linear(x, w, b, residual=r)
    else:
        ref = lambda: quick_gelu_(F.linear(x, w, b)); new = lambda: linear(x, w, b * 1.702, act="silu", alpha=1.702) ; ref0 = ref; ref = lambda: ref0() * 1.702
    t0 = time.perf_counter(); new(); torch.cuda.synchronize(); tt = time.perf_counter() - t0
    err = (ref() - new()).abs().max().item()
    tr, tn = bench(ref), bench(new)
    print(f"{name} {mode} M={M} K={K} N={N}: torch {tr*1e6:.0f} us, fused {tn*1e6:.0f} us ({2.0*M*K*N/tn/1e12:.1f} TFLOP/s), first call {tt:.1f} s, max diff {err:.2e}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
print("dumped", linear_dump_tuned("gpurun_out/linear_probe.csv"))
