"""Developer aid (GPU box): a few host->host pipelined encodes with given slots / sub-batch, for tracing under rocprofv3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lrf_amd import _lib  # noqa: E402

slots, sub = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B, H, W, RANKS = 256, 512, 768, [7, 3, 3]
g = torch.Generator().manual_seed(0)
host = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).pin_memory()
dims = _lib.plane_dims(H, W)
Uh = torch.empty((B, sum(d[4] * r for d, r in zip(dims, RANKS))), dtype=torch.int8, pin_memory=True)
Vh = torch.empty((B, 64 * sum(RANKS)), dtype=torch.int8, pin_memory=True)
pipe = _lib.Pipe(0, slots=slots, sub_batch=sub)
for _ in range(reps):
    pipe.encode_rgb_host(host, RANKS, 10, -16, 15, out=(Uh, Vh))
pipe.close()
