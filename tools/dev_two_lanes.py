"""Developer aid (GPU box): the HBM-resident encode of the bench workload split over L contexts with their own streams
(L lanes, each a contiguous share of the batch) against the one-stream encode."""
import ctypes
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lrf_amd import _lib  # noqa: E402

B, H, W, RANKS = 256, 512, 768, [7, 3, 3]
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
dims = _lib.plane_dims(H, W)
nu, nv = sum(d[4] * r for d, r in zip(dims, RANKS)), 64 * sum(RANKS)
U = torch.empty((B, nu), dtype=torch.int8, device="cuda")
V = torch.empty((B, nv), dtype=torch.int8, device="cuda")
R = (ctypes.c_int * 3)(*RANKS)
ref = _lib.context(0).encode_rgb(imgs, RANKS, 10, -16, 15)
torch.cuda.synchronize()
for lanes in (1, 2, 3, 4):
    ctxs = [_lib.Context(0) for _ in range(lanes)]  # own non-blocking streams
    per = (B + lanes - 1) // lanes

    def run():
        for i, c in enumerate(ctxs):
            b0, b1 = i * per, min(B, (i + 1) * per)
            _lib.check(lib.lrf_qmf_encode_rgb_u8(c._h, ctypes.c_void_p(imgs[b0:b1].data_ptr()), b1 - b0, H, W, R, 10, -16, 15, None,
                                                 ctypes.c_void_p(U[b0:b1].data_ptr()), ctypes.c_void_p(V[b0:b1].data_ptr())))
        for c in ctxs:
            c.synchronize()

    for _ in range(4):
        run()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        run()
    dt = (time.perf_counter() - t0) / n
    ok = torch.equal(U, ref[0]) and torch.equal(V, ref[1])
    print(f"lanes {lanes}: {dt * 1e3:.3f} ms per batch ({B * H * W / dt / 1e9:.2f} Gpix/s) same={ok}", flush=True)
    for c in ctxs:
        c.close()
