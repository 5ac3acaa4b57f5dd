"""Diagnostic: device time of qmf_decode's kernel for a batch (default 256 x 512x768) across rank triples."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lrf_amd import _lib
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 512, 768)
ctx = _lib.context(0)
dims = _lib.plane_dims(H, W)
g = torch.Generator(device="cuda").manual_seed(0)
for ranks in ((4, 2, 2), (7, 3, 3), (8, 4, 4), (8, 8, 8), (10, 5, 5), (16, 8, 8), (20, 10, 10), (26, 13, 13)):
    U = torch.randint(-16, 16, (B, sum(d[4] * r for d, r in zip(dims, ranks))), dtype=torch.int8, device="cuda", generator=g)
    V = torch.randint(-16, 16, (B, 64 * sum(ranks)), dtype=torch.int8, device="cuda", generator=g)
    for _ in range(3): out = ctx.decode_rgb(U, V, H, W, ranks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = ctx.decode_rgb(U, V, H, W, ranks)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"ranks {ranks}: {dt*1e3:.3f} ms per {B} x {H}x{W} = {B*H*W/dt/1e9:.0f} Gpixel/s", flush=True)
