"""Diagnostic: BCD kernel time per launch by iteration count at given ranks (first iteration = ordered chain, later ones = exact solve)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[2])
import torch, lrf_amd
RANKS = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (20, 10, 10)
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (64, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
prev = 0.0
for K in (1, 2, 3, 10):
    for _ in range(2): lrf_amd.qmf_factorize_batch(imgs, RANKS, num_iters=K)
    torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    for _ in range(3): lrf_amd.qmf_factorize_batch(imgs, RANKS, num_iters=K)
    torch.cuda.synchronize(); ctx.profile(False)
    b = ctx.kernel_time(_lib.LRF_K_BCD)[0] / 3; v = ctx.kernel_time(_lib.LRF_K_VUPDATE)[0] / 3
    print(f"ranks {RANKS} K={K}: k_bcd total {b:.4f} ms (+{b - prev:.4f}), k_vupdate total {v:.4f} ms")
    prev = b
