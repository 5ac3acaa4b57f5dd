"""Diagnostic: time k_bcd with parts removed (ablation builds; outputs are wrong by construction)."""
import os, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "lrf_amd", sys.argv[1])
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (256, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
print(sys.argv[1], {n: round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 4) for k, n in _lib.KERNEL_NAMES.items()})
