"""Diagnostic: the any-shape SVD initialisation of the library (lrf_qmf_svd_init_f32: k_any_gram, k_any_tridiag_reg /
k_any_eig, k_any_prod, k_any_signfix) against its restatement in oracle/lrf_oracle_any.c, bit for bit, over the three
tridiagonalisation variants (n <= 64 / 193..256, 65..192, > 256), tall and wide matrices, ranks up to min(M, N) and above,
rank-deficient and constant matrices, with and without sign vectors."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
oracle.build()
ctx = _lib.context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
shapes = [(600, 16, 3), (96, 64, 5), (200, 17, 17), (50, 100, 7), (300, 192, 5), (500, 100, 30), (128, 128, 40), (70, 300, 20), (400, 65, 9),
          (200, 256, 20), (230, 200, 33), (96, 256, 70), (400, 300, 9), (300, 520, 25), (1100, 600, 12), (40, 40, 45), (5, 9, 3), (9, 5, 6),
          (1, 7, 1), (7, 1, 1), (2, 2, 2), (3, 100, 2), (64, 1024, 8)]
shapes += [(int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.integers(1, 40))) for _ in range(25)]
if len(sys.argv) > 2:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]
bad = 0
for (M, N, R) in shapes:
    for kind in ("smooth", "u8", "rankdef"):
        if kind == "smooth":
            X = (rng.random((M, N)) * 255).astype(np.float32)
        elif kind == "u8":
            X = rng.integers(0, 256, (M, N)).astype(np.float32)
        else:
            k = max(1, min(M, N) // 3)
            X = (rng.integers(0, 16, (M, k)) @ rng.integers(0, 16, (k, N))).astype(np.float32)
        sign = None if kind == "smooth" else (rng.integers(0, 2, R) * 2 - 1).astype(np.int8)
        try:
            u, v = ctx.svd_init(torch.from_numpy(X[None]).cuda(), R, None if sign is None else torch.from_numpy(sign[None]).cuda())
        except (ValueError, NotImplementedError) as e:
            print(f"{M}x{N} R={R} {kind}: rejected ({str(e)[:60]})"); continue
        u, v = u[0].cpu().numpy(), v[0].cpu().numpy()
        uo, vo = oracle.svd_topr_any(X, R, sign)
        ok = np.array_equal(u.view(np.int32), uo.view(np.int32)) and np.array_equal(v.view(np.int32), vo.view(np.int32))
        if not ok:
            bad += 1
            du = np.abs(u - uo).max() / max(np.abs(uo).max(), 1e-30); dv = np.abs(v - vo).max() / max(np.abs(vo).max(), 1e-30)
            print(f"{M}x{N} R={R} {kind}: MISMATCH  rel du {du:.2e} dv {dv:.2e}  differing u {int((u != uo).sum())}/{u.size} v {int((v != vo).sum())}/{v.size}", flush=True)
            cols = [r for r in range(R) if not (np.array_equal(u[:, r], uo[:, r]) and np.array_equal(v[:, r], vo[:, r]))]
            print("   columns", cols, " |v_r| gpu", [float(f"{np.linalg.norm(v[:, r]):.3e}") for r in cols][:8], " oracle", [float(f"{np.linalg.norm(vo[:, r]):.3e}") for r in cols][:8],
                  " |u_r|", [float(f"{np.linalg.norm(u[:, r]):.3e}") for r in cols][:8], [float(f"{np.linalg.norm(uo[:, r]):.3e}") for r in cols][:8])
            r = cols[0]; i = int(np.argmax(np.abs(v[:, r] - vo[:, r]))); j = int(np.argmax(np.abs(u[:, r] - uo[:, r])))
            print(f"   first column {r}: v[{i}] gpu {v[i, r]!r} oracle {vo[i, r]!r}; u[{j}] gpu {u[j, r]!r} oracle {uo[j, r]!r}")
        else:
            print(f"{M}x{N} R={R} {kind}: ok", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
