"""Fuzz of the any-shape path against the oracle (development aid): random shapes, ranks and bounds, the library's own
initialisation and K iterations, bit for bit on the int8 factors and on the fp32 initial factors.
python tools/dev_fuzz_anyshape.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.context(0)
bad = 0
t0 = time.time()
for case in range(cases):
    kind = rng.integers(0, 5)
    if kind == 0:   M, N = int(rng.integers(1, 80)), int(rng.integers(1, 80))
    elif kind == 1: M, N = int(rng.integers(190, 330)), int(rng.integers(190, 330))      # around the 192 / 256 switches
    elif kind == 2: M, N = int(rng.integers(300, 700)), int(rng.integers(16, 70))        # tall, thin
    elif kind == 3: M, N = int(rng.integers(8, 40)), int(rng.integers(300, 1200))        # wide
    else:           M, N = int(rng.integers(380, 560)), int(rng.integers(380, 560))      # around 512
    if N == 64:
        N = 65  # 64 columns with R <= 64 belong to the tuned kernels (another initialisation, another oracle)
    R = int(rng.integers(1, max(2, min(M, N, 70)) + 1))
    if rng.integers(0, 6) == 0:
        R = int(min(M, N) + rng.integers(0, 4))       # ranks at / beyond the side
    R = max(1, min(R, 120))
    K = int(rng.integers(1, 4))
    lo, hi = [(-16, 15), (-128, 127), (-8, 7)][int(rng.integers(0, 3))]
    style = rng.integers(0, 3)
    if style == 0:
        k = max(1, min(M, N) // 3)
        X = (rng.integers(0, 12, (M, k)) @ rng.integers(0, 12, (k, N))).astype(np.float32)   # rank deficient, integer valued
    elif style == 1:
        X = (rng.random((M, N)) * 255).astype(np.float32)
    else:
        base = rng.normal(size=(M, 6)) @ rng.normal(size=(6, N)) * 30 + 120
        X = np.clip(base + rng.normal(size=(M, N)) * 3, 0, 255).astype(np.float32)
    # a batch of three matrices of the shape (the second a transformed copy, the third a constant), each checked on its own
    Xs = np.stack([X, np.ascontiguousarray(X[::-1, ::-1]) * np.float32(0.5) + np.float32(3), np.full_like(X, 7)])
    Xd = torch.from_numpy(Xs).cuda()
    u0, v0 = ctx.svd_init(Xd, R)
    U, V = ctx.decompose(Xd, R, K, lo, hi)
    for b in range(3):
        uo, vo = oracle.svd_topr_any(Xs[b], R)
        ok_init = np.array_equal(u0[b].cpu().numpy().view(np.int32), uo.view(np.int32)) and np.array_equal(v0[b].cpu().numpy().view(np.int32), vo.view(np.int32))
        ub, vb = oracle.bcd(Xs[b], uo, vo, K, (lo, hi))
        ok_bcd = np.array_equal(U[b].cpu().numpy(), ub.astype(np.int8)) and np.array_equal(V[b].cpu().numpy(), vb.astype(np.int8))
        if not (ok_init and ok_bcd):
            bad += 1
            print(f"MISMATCH case {case} matrix {b}: M={M} N={N} R={R} K={K} bounds=({lo},{hi}) style={style} init_ok={ok_init} bcd_ok={ok_bcd}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
