"""Diagnostic: k_init time when stopped after stage n (LRF_DEBUG_INIT_SWEEPS=n: 1 Gram, 2 tridiagonalisation,
3 eigenvalues, 0 everything).  Run one process per stage (the variable is read when the context is created)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import _lib
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
ms, n = ctx.kernel_time(_lib.LRF_K_INIT if hasattr(_lib, "LRF_K_INIT") else 1)
print("stage", os.environ.get("LRF_DEBUG_INIT_SWEEPS", "0"), {nm: round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 4) for k, nm in _lib.KERNEL_NAMES.items() if nm == "k_init"})
