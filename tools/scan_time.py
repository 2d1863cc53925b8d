"""Time one flat-index search on the GPU: python tools/scan_time.py NQ N D K [algo] (tuning aid)."""
import sys, time, torch
import lemon_amd
from lemon_amd import IndexFlatIP, IndexFlatL2
nq, n, d, k = (int(v) for v in sys.argv[1:5])
algo = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=1)
q = torch.nn.functional.normalize(torch.randn(nq, d, device=dev, generator=g), dim=1)
import os
idx = (IndexFlatL2 if os.environ.get("METRIC", "ip") == "l2" else IndexFlatIP)(d); idx.set_algo({"auto": 0, "f32": 1, "bf16": 2}[algo]); idx.add(x)
idx.search(q, k); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); idx.search(q, k); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
D, I = idx.search(q, k); torch.cuda.synchronize()
w = torch.arange(1, k + 1, device=dev, dtype=torch.int64)       # position-weighted: an order change shows too
chk = f"I:{int(((I + 1) * w).sum()) & 0xffffffffffff:012x} D:{int((D.view(torch.int32).to(torch.int64) * w).sum()) & 0xffffffffffff:012x}"
print(f"nq={nq} n={n} d={d} k={k} algo={algo} best={t*1e3:.2f} ms  {2.0*nq*n*d/t/1e12:.1f} TFLOP/s  checksum {chk}", flush=True)
