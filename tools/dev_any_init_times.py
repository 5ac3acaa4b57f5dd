"""k_init class time (Gram + tridiagonalisation + eigen-solver + long factor) of the any-shape initialisation for the
shapes of the patch-size sweep, 256 matrices each.  python tools/dev_any_init_times.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
ctx = _lib.context(0)
for (B, M, N, R) in ((256, 1536, 256, 51), (256, 384, 256, 26), (768, 384, 256, 26), (256, 512, 768, 102), (256, 256, 384, 26), (256, 384, 1024, 77)):
    X = torch.rand(B, M, N, device="cuda") * 255
    ctx.svd_init(X, R)
    torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    ctx.svd_init(X, R); torch.cuda.synchronize()
    ctx.profile(False)
    print(f"B,M,N,R={(B, M, N, R)}: init {ctx.kernel_time(_lib.LRF_K_INIT)[0]:.2f} ms", flush=True)
