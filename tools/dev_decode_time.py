"""Diagnostic: k_decode throughput at the bench workload (256 x 512x768, ranks (7,3,3))."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
for _ in range(3): out = ctx.decode_rgb(U, V, 512, 768, (7, 3, 3))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = ctx.decode_rgb(U, V, 512, 768, (7, 3, 3))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
px = 256 * 512 * 768
print(f"k_decode: {ms:.4f} ms per 256 images = {px/ms/1e6:.0f} Gpix/s; output {3*px/1e6:.0f} MB + factors {(U.numel()+V.numel())/1e6:.1f} MB -> {(3*px+U.numel()+V.numel())/ms/1e9:.2f} TB/s")
