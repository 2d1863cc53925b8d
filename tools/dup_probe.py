"""fp32 scan vs bf16 filter scan on class-prompt style data (exact duplicates) at multi-GPU DB sizes (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lemon_amd as hip
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
d, k, C = 512, 51, 100
proto = hip.normalize_vectors(torch.randn(C, d, device=dev, generator=g) * 0.05 + torch.randn(1, d, device=dev, generator=g))  # close prompts (cos ~0.97+)
print("min cos between prototypes:", (proto @ proto.T).min().item())
for n in (40000, 160000, 320000):
    X = proto[torch.randint(0, C, (n,), device=dev, generator=g)].contiguous()
    Q = proto[torch.randint(0, C, (50000,), device=dev, generator=g)].contiguous()
    res = {}
    for name, algo in (("f32", 1), ("bf16", 2)):
        idx = hip.IndexFlatIP(d); idx.set_algo(algo); idx.add(X)
        idx.search(Q, k); torch.cuda.synchronize()
        t0 = time.perf_counter(); D, I = idx.search(Q, k); torch.cuda.synchronize(); t = time.perf_counter() - t0
        res[name] = (t, D, I)
    same = torch.equal(res["f32"][1], res["bf16"][1]) and torch.equal(res["f32"][2], res["bf16"][2])
    print(f"n={n}: f32 {res['f32'][0]*1e3:.1f} ms, bf16 {res['bf16'][0]*1e3:.1f} ms, identical={same}", flush=True)
