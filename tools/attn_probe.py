"""Fused attention vs torch SDPA on the encoder shapes (tuning aid)."""
import time, torch, torch.nn.functional as F
from lemon_amd.ops import attention
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for B, L, H, causal in [(1000, 50, 12, False), (1000, 8, 8, True), (1000, 77, 8, True), (256, 197, 12, False), (128, 257, 16, False)]:
    W = 64 * H
    qkv = torch.randn(B, L, 3 * W, device=dev)
    def sdpa():
        q, k, v = qkv.view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(B, L, W)
    t_ref, t_new = bench(sdpa), bench(lambda: attention(qkv, H, causal))
    err = (sdpa() - attention(qkv, H, causal)).abs().max().item()
    gb = 16.0 * B * L * W / 1e9
    print(f"B={B} L={L} H={H} causal={causal}: sdpa+copy {t_ref*1e6:.0f} us, fused {t_new*1e6:.0f} us ({gb/t_new:.0f} GB/s), max diff {err:.2e}", flush=True)
