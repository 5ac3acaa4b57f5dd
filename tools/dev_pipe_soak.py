import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from lrf_amd import _lib
B, H, W, RANKS = 100, 512, 768, [7, 3, 3]
g = torch.Generator().manual_seed(1)
host = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
dims = _lib.plane_dims(H, W)
def outs():
    return (torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8), torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8))
pipe = _lib.Pipe(0, slots=2, sub_batch=16)
U0, V0 = outs()
pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(U0, V0))
bad = 0
t0 = time.perf_counter()
for it in range(150):
    U, V = outs()
    pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(U, V))
    if not (torch.equal(U, U0) and torch.equal(V, V0)): bad += 1
print(f"150 pageable pipelined encodes of {B} images (7 sub-batches, 2 threads): {bad} differ, {time.perf_counter()-t0:.1f} s")
pipe3 = _lib.Pipe(0, slots=3, sub_batch=8)
for it in range(50):
    U, V = outs()
    pipe3.encode_rgb_host(host, RANKS, 10, -16, 15, out=(U, V))
    if not (torch.equal(U, U0) and torch.equal(V, V0)): bad += 1
print("3 slots x 8 images:", bad, "differ in total")
sys.exit(1 if bad else 0)
