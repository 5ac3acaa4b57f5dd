"""constant / rank-one matrices through the any-shape initialisation against the oracle (development aid)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
ctx = _lib.context(0)
for (M, N, R) in ((47, 29, 28), (29, 311, 21), (308, 325, 25), (474, 396, 120), (100, 90, 10), (100, 90, 1), (100, 90, 2), (300, 200, 5)):
    X = np.full((M, N), 7, np.float32)
    u0, v0 = ctx.svd_init(torch.from_numpy(X[None]).cuda(), R)
    uo, vo = oracle.svd_topr_any(X, R)
    u0, v0 = u0[0].cpu().numpy(), v0[0].cpu().numpy()
    du = (u0.view(np.int32) != uo.view(np.int32)); dv = (v0.view(np.int32) != vo.view(np.int32))
    cols = sorted(set(np.nonzero(du.any(0))[0]) | set(np.nonzero(dv.any(0))[0]))
    print(f"M={M} N={N} R={R}: differing columns {cols[:12]}{'...' if len(cols) > 12 else ''}; max |u0| per column (first 4) {np.abs(u0).max(0)[:4]}, oracle {np.abs(uo).max(0)[:4]}; "
          f"max abs diff {max(np.abs(u0 - uo).max(), np.abs(v0 - vo).max()):.3e}", flush=True)
