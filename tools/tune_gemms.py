"""Regenerate lemon_amd/data/tunableop_gfx950.csv on an MI355X: one pass of the headline encoder
workload with TunableOp tuning every GEMM shape it meets.  Run via gpurun; writes gpurun_out/tunableop_gfx950.csv."""
import os, shutil, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd.tuning import enable_gemm_tuning
from lemon_amd.clip import ClipConfig, LemonCLIP
from lemon_amd import datasets as ds
from lemon_amd.clip import SyntheticTokenizer

archs = sys.argv[1:] or ["vit-b-32"]
work = enable_gemm_tuning(tune_missing=True, max_ms=int(os.environ.get("TUNE_MS", "4000")), max_iters=50, results=None)
dev = torch.device("cuda:0")
for arch in archs:
    cfg = ClipConfig.named(arch)
    model = LemonCLIP(cfg).eval().to(dev)
    tok = SyntheticTokenizer(cfg.vocab_size, cfg.context_length, cfg.eos_token_id)
    ids = torch.tensor(tok(["A photo of a " + l for l in ds.cifar100_labels] * 10, padding="max_length", truncation=True)["input_ids"]).to(dev)
    bs = {"vit-b-32": 1000, "vit-b-16": 256, "vit-l-14": 128}.get(arch, 256)
    px = torch.randn(bs, 3, cfg.image_size, cfg.image_size, device=dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        model.encode_image(px); model.encode_text(ids[:bs])
    torch.cuda.synchronize()
    print(arch, "tuned in", round(time.perf_counter() - t0, 1), "s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
import torch.cuda.tunable as tn
print(len(tn.get_results()), "results")
with open("gpurun_out/tunableop_gfx950.csv", "w") as f:
    for k, v in tn.get_validators():
        f.write(f"Validator,{k},{v}\n")
    for r in tn.get_results():
        f.write(",".join(str(x) for x in r) + "\n")
if os.path.exists(work):
    shutil.copyfile(work, "gpurun_out/tunableop_gfx950_raw.csv")
