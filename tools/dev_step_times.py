import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(1234)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
dims = _lib.plane_dims(512, 768)
U = torch.empty((256, sum(d[4] * r for d, r in zip(dims, (7, 3, 3)))), dtype=torch.int8, device="cuda")
V = torch.empty((256, 64 * 13), dtype=torch.int8, device="cuda")
ts = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3), out=(U, V))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.3f}" for t in ts))
