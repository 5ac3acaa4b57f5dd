import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lrf_amd import _lib
ctx = _lib.context(0)
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(2): ctx.svd_encode_rgb(imgs, 5)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): U, V, qp = ctx.svd_encode_rgb(imgs, 5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print(f"svd_encode 256 x 512x768x3, R=5: {dt*1e3:.2f} ms per batch = {256*512*768/dt/1e9:.2f} Gpix/s")
