"""Encoder GEMM shapes: stock hipBLASLt heuristic vs TunableOp (tuning aid).  python tools/gemm_probe.py [M]"""
import os, sys, time, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda:0")
shapes = [("qkv", 768, 2304), ("out", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]
def run(tag):
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        for _ in range(3): torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        print(f"{tag} {name} M={M} K={K} N={N}: {t*1e6:.0f} us  {2.0*M*K*N/t/1e12:.1f} TFLOP/s", flush=True)
run("stock")
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "3000"))); tn.set_max_tuning_iterations(20)
tn.set_filename("gpurun_out/tunableop_probe.csv")
t0 = time.perf_counter(); run("tuned"); print("tuning+run s:", time.perf_counter() - t0)
tn.write_file()
