"""Does the planted synthetic CIFAR-like set carry class signal through a random-init CLIP? (diagnostic)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lemon_amd.clip import ClipConfig, LemonCLIP
from lemon_amd.pipeline import Embedder
from lemon_amd import IndexFlatIP
args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
sys.argv = ["bench.py", "--n_train", "6000", "--n_val", "500", "--n_test", "500"]
args = bench.parse()
dev = torch.device("cuda:0")
cfg = ClipConfig.named("vit-b-32"); model = LemonCLIP(cfg)
data = bench.make_cifar_like(args, cfg, 0, dev)
emb = Embedder(model, dev, batch_size=1000)
e_img = emb.embed_images(data["train"]["pixels"]); e_txt = emb.embed_texts(data["train"]["ids"])
print("img emb std over samples:", e_img.std(0).mean().item(), " txt:", e_txt.std(0).mean().item())
idx = IndexFlatIP(cfg.embed_dim); idx.add(e_img)
D, I = idx.search(e_img, 11)
clean = torch.from_numpy(data["train"]["clean"]).to(dev)
agree = (clean[I[:, 1:]] == clean[:, None]).float().mean().item()
print("image-kNN clean-class agreement (k=10):", agree, " D range", D[:, 1].min().item(), D[:, 1].max().item())
sim = (e_txt @ e_txt.T)
print("text cos between different prompts: min", sim.min().item(), "mean", sim.mean().item())
