"""Diagnostic: per-kernel times of the bench workload for a given build of the library (argv[1] = file under lrf_amd/)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import torch, lrf_amd
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
for _ in range(5): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
print(sys.argv[1], {n: round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 4) for k, n in _lib.KERNEL_NAMES.items()})
