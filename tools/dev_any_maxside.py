"""The largest side the any-shape initialisation accepts (min(M, N) = 2048) against the oracle, bit for bit (development aid;
the oracle needs a few minutes for it)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
rng = np.random.default_rng(0)
for (M, N, R) in ((2048, 2060, 5), (1500, 1400, 7)):
    base = rng.normal(size=(M, 8)) @ rng.normal(size=(8, N)) * 20 + 100
    X = np.clip(base + rng.normal(size=(M, N)) * 4, 0, 255).astype(np.float32)
    ctx = _lib.context(0)
    t0 = time.time()
    u0, v0 = ctx.svd_init(torch.from_numpy(X[None]).cuda(), R); torch.cuda.synchronize()
    t1 = time.time()
    uo, vo = oracle.svd_topr_any(X, R)
    t2 = time.time()
    ok = np.array_equal(u0[0].cpu().numpy().view(np.int32), uo.view(np.int32)) and np.array_equal(v0[0].cpu().numpy().view(np.int32), vo.view(np.int32))
    print(f"{M}x{N} R={R}: GPU {t1 - t0:.2f} s, oracle {t2 - t1:.1f} s, bit-identical {ok}", flush=True)
