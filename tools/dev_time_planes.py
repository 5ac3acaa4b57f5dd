"""Diagnostic: time of the patch-matrix pass (K1) on large batches: B x HxW from argv (default 512 x 1365x2048)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lrf_amd import _lib
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 1365, 2048)
ctx = _lib.context(0)
g = torch.Generator(device="cuda").manual_seed(1)
imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(3): X = ctx.planes_from_rgb(imgs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): X = ctx.planes_from_rgb(imgs)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
alg = imgs.numel() + X.numel() * 4
print(f"{B} x {H}x{W}: {dt*1e3:.3f} ms, {alg/dt/1e12:.2f} TB/s algorithmic ({alg/1e9:.2f} GB)")
