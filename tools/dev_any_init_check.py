import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lrf_amd import _lib
ctx = _lib.context(0)
rng = np.random.default_rng(3)
for (M, N, R) in ((96, 256, 70), (300, 192, 12), (500, 100, 30), (128, 128, 40), (70, 300, 20), (400, 65, 9)):
    X = (rng.random((2, M, N)) * 255).astype(np.float32)
    u, v = ctx.svd_init(torch.from_numpy(X).cuda(), R)
    u, v = u.cpu().numpy().astype(np.float64), v.cpu().numpy().astype(np.float64)
    for b in range(2):
        s = np.linalg.svd(X[b].astype(np.float64), compute_uv=False)[:R]
        rec = u[b] @ v[b].T
        Xd = X[b].astype(np.float64)
        # best rank-R error
        U, S, Vt = np.linalg.svd(Xd, full_matrices=False)
        best = np.linalg.norm(Xd - (U[:, :R] * S[:R]) @ Vt[:R])
        got = np.linalg.norm(Xd - rec)
        sv = np.sqrt(np.linalg.norm(u[b], axis=0) * np.linalg.norm(v[b], axis=0))  # sqrt(s)*sqrt(s) norms -> s
        print(M, N, R, b, "err/best %.6f" % (got / best), "max rel sv err %.2e" % np.max(np.abs(np.linalg.norm(u[b], axis=0) * np.linalg.norm(v[b], axis=0) - s) / s[0]))
