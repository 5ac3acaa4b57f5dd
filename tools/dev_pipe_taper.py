"""Developer aid (GPU box): host->host pipelined encode at the bench workload (or `clic`) for several tail schedules and slot
counts (LRF_PIPE_BULK / LRF_PIPE_TAIL are read per submission)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lrf_amd import _lib  # noqa: E402
clic = len(sys.argv) > 1 and sys.argv[1] == "clic"
B, H, W, RANKS = (512, 1365, 2048, [7, 3, 3]) if clic else (256, 512, 768, [7, 3, 3])
g = torch.Generator().manual_seed(0)
host = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).pin_memory()
dims = _lib.plane_dims(H, W)
Uh = torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8, pin_memory=True)
Vh = torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8, pin_memory=True)
dev = torch.empty((B, 3, H, W), dtype=torch.uint8, device="cuda")
for _ in range(3): dev.copy_(host, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dev.copy_(host, non_blocking=True)
torch.cuda.synchronize(); t_copy = (time.perf_counter() - t0) / 5
print(f"H2D alone: {t_copy * 1e3:.3f} ms"); del dev
if clic:
    scheds = [("", ""), ("32", "24,8"), ("32", "16,8,4"), ("16", "12,4"), ("64", "32,16,8"), ("32", "32"), ("8", "8")]
else:
    scheds = [("", ""), ("32", "32"), ("32", "24,8"), ("32", "16,8"), ("32", "8,8"), ("32", "16,16"), ("32", "16,8,8"), ("48", "24,8"), ("32", "28,4"), ("64", "48,16"), ("64", "16,8")]
    if len(sys.argv) > 2: scheds = [("32", "32"), ("32", "24,8"), ("64", "48,16")]
for slots in (2,):
    for bulk, tail in scheds:
        for k, v in (("LRF_PIPE_BULK", bulk), ("LRF_PIPE_TAIL", tail)):
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        pipe = _lib.Pipe(0, slots=slots, sub_batch=0)
        for _ in range(3): pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(Uh, Vh))
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(Uh, Vh)); ts.append(time.perf_counter() - t0)
        ts.sort()
        sizes = [n for _, n, _, _ in pipe.encode_rgb_host_iter(host, RANKS, 10, -16, 15, out=(Uh, Vh))]
        print(f"slots {slots} bulk {bulk or 'auto':>4s} tail {tail or 'auto':>9s}: median {ts[3]*1e3:7.3f} ms min {ts[0]*1e3:7.3f}  frac of copy {t_copy/ts[3]:.3f}  pieces {sizes[:3]}..{sizes[-4:]}", flush=True)
        pipe.close()
