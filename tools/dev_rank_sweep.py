"""Diagnostic: ms per batch of 64 x 512x768 images (LRF_SWEEP_BATCH overrides the 64) for rank triples across the three BCD
kernel families."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
NB = int(os.environ.get("LRF_SWEEP_BATCH", "64"))
imgs = torch.randint(0, 256, (NB, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
TRIPLES = ((4, 2, 2), (7, 3, 3), (8, 4, 4), (10, 5, 5), (16, 8, 8), (20, 10, 10), (26, 13, 13), (40, 20, 20))
if len(sys.argv) > 1:  # e.g. "26,13,13": one triple only (for a rocprofv3 --stats run)
    TRIPLES = (tuple(int(v) for v in sys.argv[1].split(",")),)
for ranks in TRIPLES:
    for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()  # timed with kernel profiling OFF (with it on, the kernel families of a call do not run side by side)
    for _ in range(5): lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ctx.profile(True); ctx.profile_reset()
    for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, ranks)
    torch.cuda.synchronize()
    ctx.profile(False)
    k = {nm: round(ctx.kernel_time(i)[0] / 3, 3) for i, nm in _lib.KERNEL_NAMES.items() if ctx.kernel_time(i)[1]}
    print(f"ranks {ranks}: {dt*1e3:.2f} ms per {NB} images = {NB*512*768/dt/1e9:.1f} Gpix/s  {k}")
