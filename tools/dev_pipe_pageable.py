import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from lrf_amd import _lib
B, H, W, RANKS = 256, 512, 768, [7, 3, 3]
g = torch.Generator().manual_seed(0)
host = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
pinned = host.pin_memory()
dims = _lib.plane_dims(H, W)
Uh = torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8, pin_memory=True)
Vh = torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8, pin_memory=True)
pipe = _lib.Pipe(0, slots=2, sub_batch=32)
for name, src in (("pinned", pinned), ("pageable", host)):
    for _ in range(3): pipe.encode_rgb_host(src, RANKS, 10, -16, 15, out=(Uh, Vh))
    t0 = time.perf_counter()
    for _ in range(5): pipe.encode_rgb_host(src, RANKS, 10, -16, 15, out=(Uh, Vh))
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms per 256 images ({B*H*W/dt/1e9:.2f} Gpix/s)")
dev = torch.empty_like(host, device="cuda")
for _ in range(2): dev.copy_(host)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dev.copy_(host)
torch.cuda.synchronize(); print(f"plain pageable H2D: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(5): pinned.copy_(host)
print(f"host memcpy pageable -> pinned (one thread): {(time.perf_counter()-t0)/5*1e3:.2f} ms")
