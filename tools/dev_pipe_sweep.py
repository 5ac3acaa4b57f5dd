"""Developer aid (GPU box): host->host pipelined encode, sweep of slots x sub-batch at the bench workload, with the
arrival time of every sub-batch on the host."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lrf_amd import _lib  # noqa: E402

B, H, W, RANKS = 256, 512, 768, [7, 3, 3]
g = torch.Generator().manual_seed(0)
host = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).pin_memory()
dims = _lib.plane_dims(H, W)
Uh = torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8, pin_memory=True)
Vh = torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8, pin_memory=True)
dev = torch.empty((B, 3, H, W), dtype=torch.uint8, device="cuda")
for _ in range(3):
    dev.copy_(host, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(host, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D alone: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
combos = [(s, sb) for s in (1, 2, 3, 4, 6) for sb in (8, 16, 32, 64, 128)]
if len(sys.argv) > 1:
    combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for slots, sub in combos:
    pipe = _lib.Pipe(0, slots=slots, sub_batch=sub)
    for _ in range(3):
        pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(Uh, Vh))
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(Uh, Vh))
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    arr = []
    for first, cnt, _, _ in pipe.encode_rgb_host_iter(host, RANKS, 10, -16, 15, out=(Uh, Vh)):
        arr.append((time.perf_counter() - t0) * 1e3)
    t_submit = arr[0] if arr else 0
    print(f"slots {slots} sub {sub:4d}: {dt * 1e3:7.3f} ms  ({B * H * W / dt / 1e9:5.2f} Gpix/s)  arrivals ms: "
          + " ".join(f"{a:.2f}" for a in arr[:12]), flush=True)
    pipe.close()
