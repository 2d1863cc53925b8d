"""Time the attention kernels at the tower shapes: python tools/attn_time.py  (old = first general kernel, new = staged split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemon_amd import _lib, ops
if os.environ.get("DBG_SO"): _lib.SO_PATH = os.environ["DBG_SO"]
lib = _lib.load()
for B, L, H in ((256, 197, 12), (128, 257, 16), (332, 197, 12), (2620, 50, 12), (64, 77, 8)):
    qkv = torch.randn(B, L, 3 * H * 64, device="cuda")
    res = {}
    for name, mode in (("new", 1), ("old", 2), ("new", 1), ("old", 2)):
        lib.lemon_attention_set_f16(mode)
        for fn_name, fn in (("f32out", lambda: ops.attention(qkv, H, False)), ("tiled", lambda: ops.attention_t(qkv, H, False))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault((name, fn_name), []).append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.lemon_attention_set_f16(1)
    gb = B * L * H * 64 * 4 * 4 / 1e9
    print(f"B={B} L={L} H={H}: " + "  ".join(f"{k[0]}/{k[1]} {min(v):.1f} us ({gb / (min(v) * 1e-6) / 1e3:.2f} TB/s)" for k, v in res.items()), flush=True)
