"""Diagnostic: k_init time for a given build of the library (argv[1] = file name under lrf_amd/), optionally stopped
after a stage (LRF_DEBUG_INIT_SWEEPS=1 Gram, 2 tridiagonalisation, 3 eigenvalues).  The round-1 Gram ablations
(loads replaced by constants / MFMAs removed) were throw-away builds timed with this script."""
import os, sys
sys.path.insert(0, "/root/repo")
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join("/root/repo", "lrf_amd", sys.argv[1])
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
print(sys.argv[1], os.environ.get("LRF_DEBUG_INIT_SWEEPS"), {n: round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 4) for k, n in _lib.KERNEL_NAMES.items() if n == "k_init"})
