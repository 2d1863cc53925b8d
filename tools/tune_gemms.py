"""Regenerate lemon_amd/data/linear_gfx950.csv on an MI355X: one pass of the encoder workload with
lemon_linear_f32 in tuning mode (benchmarking every GEMM key it meets), then dump the winners with the
hipBLASLt-version / arch stamp the loader checks.
Run via gpurun: python tools/tune_gemms.py [arch[:batch[:text_batch]] ...] -> gpurun_out/linear_gfx950.csv
(text_batch defaults to 4 x batch, pipeline.Embedder's default; bench.py uses gcd(split sizes) = 5000 at the headline shape)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LEMON_LINEAR_TUNE_MS", "6000")
os.environ["LEMON_LINEAR_TUNED"] = ""          # start from scratch
os.environ["LEMON_LINEAR_TUNE"] = "1"          # the explicit OFFLINE tuning mode (never on in the inference path)
from lemon_amd import datasets as ds
from lemon_amd.clip import ClipConfig, LemonCLIP, SyntheticTokenizer
from lemon_amd.ops import linear_dump_tuned

dev = torch.device("cuda:0")
for spec in (sys.argv[1:] or ["vit-b-32:1000"]):
    arch, _, rest = spec.partition(":")
    bs, _, tbs = rest.partition(":")
    cfg = ClipConfig.named(arch)
    bs = int(bs or 256)
    tbs = int(tbs or 4 * bs)
    model = LemonCLIP(cfg).eval().to(dev)
    tok = SyntheticTokenizer(cfg.vocab_size, cfg.context_length, cfg.eos_token_id)
    prompts = (["A photo of a " + l for l in ds.cifar100_labels] * (tbs // 100 + 1))[:tbs]     # one text micro-batch
    ids = torch.tensor(tok(prompts, padding="max_length", truncation=True)["input_ids"]).to(dev)
    from lemon_amd.data import gpu_transform_batch
    u8 = torch.randint(0, 256, (bs, 32, 32, 3), dtype=torch.uint8, device=dev)
    px = gpu_transform_batch(u8, cfg.image_size, patch=cfg.patch_size)      # patch-major: the patch embedding is a GEMM
    t0 = time.perf_counter()
    for mode in os.environ.get("TUNE_MODES", "f16x3,split,f32").split(","):          # all GEMM modes (or the ones asked for) of lemon_amd/clip.py meet their keys (bf16 split operands / fp32)
        os.environ["LEMON_GEMM"] = mode
        with torch.no_grad():
            # text tower: only position-independent solutions (identical prompts must get identical embeddings wherever they
            # sit in a micro-batch: the text-side search folds them); image tower: rows are distinct images, the fastest
            # validated solution is taken
            towers = os.environ.get("TUNE_TOWERS", "image,text").split(",")
            if "image" in towers:
                os.environ["LEMON_LINEAR_ALLOW_POSITION_DEPENDENT"] = "1"
                model.encode_image(px)
            if "text" in towers:
                os.environ["LEMON_LINEAR_ALLOW_POSITION_DEPENDENT"] = "0"
                model.encode_text(ids); model.encode_text(ids[:bs])
        torch.cuda.synchronize()
    print(spec, "tuned in", round(time.perf_counter() - t0, 1), "s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
print(linear_dump_tuned("gpurun_out/linear_gfx950.csv"), "keys written")
