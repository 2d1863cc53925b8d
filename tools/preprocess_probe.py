"""lemon_preprocess_u8 throughput on the CIFAR shape vs the PIL thread pool (diagnostic)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemon_amd.data import gpu_transform_batch, ImageLabelSet
imgs = np.random.default_rng(0).integers(0, 256, (4000, 32, 32, 3), dtype=np.uint8)
u8 = torch.from_numpy(imgs).cuda()
gpu_transform_batch(u8[:1000]); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(0, 4000, 1000): out = gpu_transform_batch(u8[i:i + 1000])
torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"GPU kernel: {4000 / t:.0f} images/s ({4000 * 602112 / t / 1e9:.0f} GB/s written)")
t0 = time.perf_counter()
for i in range(0, 4000, 1000): out = gpu_transform_batch(torch.from_numpy(imgs[i:i + 1000]).cuda())
torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"GPU incl. H2D of uint8: {4000 / t:.0f} images/s")
ds_ = ImageLabelSet(imgs[:2000], np.zeros(2000, int), np.zeros(2000, int), 224, workers=16)
t0 = time.perf_counter(); n = 0
for px, _, _ in ds_.batches(500): n += px.shape[0]
t = time.perf_counter() - t0
print(f"PIL thread pool (16 threads): {n / t:.0f} images/s")
