"""Timing of the RGB colour-space branch (qmf_encode(color_space="RGB")): B x 512x768 images, rank R, 10 iterations.
Prints Mpixel/s of the factorisation (inputs resident in HBM) and the per-kernel-class times.  Development aid; the
contract benchmark of the repository is bench.py."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(2): ctx.qmf_rgbspace_encode(imgs, R)
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
n = 3
for _ in range(n): ctx.qmf_rgbspace_encode(imgs, R)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ctx.profile(False)
k = {nm: round(ctx.kernel_time(i)[0] / n, 3) for i, nm in _lib.KERNEL_NAMES.items() if ctx.kernel_time(i)[1]}
x_bytes = B * 6144 * 192 * 4
print(f"B={B} R={R}: {dt*1e3:.2f} ms per batch, {B*512*768/dt/1e6:.0f} Mpix/s; ms per batch by class {k}; "
      f"X = {x_bytes/1e6:.0f} MB read per BCD iteration -> {x_bytes/ (k.get('k_bcd',1e9)/10*1e-3)/1e9:.0f} GB/s in the BCD kernels")
