"""Diagnostic: k_init time when stopped after stage n (LRF_DEBUG_INIT_SWEEPS=n: 1 Gram partials, 2 tridiagonalisation,
3 eigenvalues, 4 twisted factorisation, 5 Gram-Schmidt, 0 everything) for uniform rank R on 256 luma-sized matrices
(lrf_qmf_svd_init path: decompose of [256, 6144, 64]).  Run one process per stage (the variable is read when the context is
created):  for s in 1 2 3 4 5 0; do LRF_DEBUG_INIT_SWEEPS=$s python tools/dev_init_stages.py 16 26; done"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import _lib
ranks = [int(a) for a in sys.argv[1:]] or [7, 16, 26]
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((256, 6144, 64), device="cuda", generator=g) * 255
ctx = _lib.context(0)
out = {}
for R in ranks:
    for _ in range(2): ctx.svd_init(X, R)
    torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    for _ in range(3): ctx.svd_init(X, R)
    torch.cuda.synchronize()
    ms, n = ctx.kernel_time(_lib.LRF_K_INIT)
    ctx.profile(False)
    out[R] = round(ms / max(n, 1), 4)
print("stage", os.environ.get("LRF_DEBUG_INIT_SWEEPS", "0"), out, flush=True)
