"""Diagnostic: ms per bench step with and without the per-kernel HIP event pairs (lrf_ctx_profile)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
dims = _lib.plane_dims(512, 768)
U = torch.empty((256, sum(d[4] * r for d, r in zip(dims, (7, 3, 3)))), dtype=torch.int8, device="cuda")
V = torch.empty((256, 64 * 13), dtype=torch.int8, device="cuda")
for prof in (False, True, False, True):
    ctx.profile(prof); ctx.profile_reset()
    for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3), out=(U, V))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3), out=(U, V))
    torch.cuda.synchronize()
    print(f"profile={prof}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per step")
ctx.profile(False)
