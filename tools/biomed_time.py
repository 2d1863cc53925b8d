"""Throughput of the biomed_clip branch (lemon_amd/biomed.py) on one MI355X: images/s of the timm-style ViT-B/16, captions/s of
the BERT tower at a few caption lengths, and the hand-written GEMM's share / roofline inside both (HIP events around its launches).
Random weights of the published architecture; usage: python tools/biomed_time.py [batch]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lemon_amd import ops                                   # noqa: E402
from lemon_amd.biomed import BiomedCLIP                     # noqa: E402
from lemon_amd.data import gpu_transform_batch, patch_operand_supported   # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ops.gemm_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pr = ops.gemm_profile_read()
    ops.gemm_profiling(False)
    return dt, pr["kernel_ms"] / reps * 1e-3, pr["flops"] / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    m = BiomedCLIP().eval().cuda()
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (B, 64, 64, 3), dtype=torch.uint8, generator=g).cuda()
    cfg = m.cfg
    out = {"batch": B, "gemm_mode": ops.gemm_mode(), "ln_fold": ops.ln_fold_enabled()}
    with torch.no_grad():
        po = gpu_transform_batch(u8, cfg.image_size, patch=cfg.patch_size, operand=patch_operand_supported(cfg.patch_size, cfg.image_size))
        dt, gk, fl = timed(lambda: m.encode_image(po), 5)
        out["image"] = {"images_per_s": B / dt, "ms": dt * 1e3, "gemm_share": gk / dt, "gemm_frac_of_2.5PF": fl / gk / 2.5e15}
        for L in (16, 64, 128, 256):
            n = max(64, B * 197 // L // 64 * 64)
            ids = torch.randint(4, cfg.vocab_size, (n, cfg.context_length), generator=g)
            ids[:, 0], ids[:, L - 1], ids[:, L:] = 2, 3, 0
            ids = ids.cuda()
            lens = torch.full((n,), L)
            dt, gk, fl = timed(lambda: m.encode_text(ids, seq_len=L, lengths=lens), 5)
            out[f"text_L{L}"] = {"captions": n, "captions_per_s": n / dt, "tokens_per_s": n * L / dt, "ms": dt * 1e3, "gemm_share": gk / dt,
                                 "gemm_frac_of_2.5PF": fl / gk / 2.5e15}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
