"""Diagnostic: per-kernel times at ranks (20,10,10) (the k_bcd_big family) for a given build of the library
(argv[1] = file under lrf_amd/; ablation builds: -DLRF_BIG_NO_GS, -DLRF_BIG_NO_P)."""
import os, sys
sys.path.insert(0, "/root/repo")
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join("/root/repo", "lrf_amd", sys.argv[1])
import torch, lrf_amd
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (64, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
RANKS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (20, 10, 10)
for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, RANKS)
torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, RANKS)
torch.cuda.synchronize()
print(sys.argv[1], RANKS, {n: round(ctx.kernel_time(k)[0] / max(ctx.kernel_time(k)[1], 1), 4) for k, n in _lib.KERNEL_NAMES.items()})
