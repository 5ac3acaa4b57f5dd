"""Randomised sweep of the any-shape path against the oracle (development aid; the pinned cases live in tests/test_anyshape.py):
  * lrf_qmf_bcd_f32 on random [M, N], R, K, bounds — int8 factors bit for bit;
  * lrf_qmf_planes_any_u8 / lrf_qmf_decode_any_u8 on random image sizes and patch sizes — bit for bit.
usage: python tools/dev_fuzz_anyshape.py [trials] [seed]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = _lib.context(0)
bad = 0
for t in range(trials):
    M = int(rng.choice([rng.integers(1, 40), rng.integers(40, 900)]))
    N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 1100), 16, 256, 1024]))
    R = int(rng.integers(1, min(130, max(2, 2 * min(M, N)))))
    K = int(rng.integers(1, 4))
    lo, hi = [(-16, 15), (-128, 127), (-4, 3), (0, 7), (-1, 1)][int(rng.integers(0, 5))]
    kind = int(rng.integers(0, 3))
    X = (rng.random((1, M, N)) * 255).astype(np.float32)
    if kind == 1: X = np.round(X)
    if kind == 2: X = (rng.normal(size=(1, M, 3)) @ rng.normal(size=(1, 3, N)) * 30 + 120).astype(np.float32)
    U0 = (rng.normal(size=(1, M, R)) * 4).astype(np.float32)
    V0 = (rng.normal(size=(1, N, R)) * 4).astype(np.float32)
    U, V = ctx.bcd(torch.from_numpy(X).cuda(), torch.from_numpy(U0).cuda(), torch.from_numpy(V0).cuda(), K, lo, hi)
    u, v = oracle.bcd(X[0], U0[0], V0[0], K, (lo, hi))
    du = int((U[0].cpu().numpy() != u.astype(np.int8)).sum()); dv = int((V[0].cpu().numpy() != v.astype(np.int8)).sum())
    if du or dv:
        bad += 1
        print(f"BCD MISMATCH M={M} N={N} R={R} K={K} bounds=({lo},{hi}) kind={kind}: U {du} V {dv}", flush=True)
print(f"bcd: {trials} trials, {bad} mismatching", flush=True)
bad2 = 0
for t in range(trials):
    H, W = int(rng.integers(8, 150)), int(rng.integers(8, 200))
    ps = [None, (4, 4), (16, 16), (32, 32), (8, 4), (2, 16), (5, 3)][int(rng.integers(0, 7))]
    img = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    try:
        dims = _lib.plane_dims_any(H, W, ps)
    except ValueError:
        continue
    want = oracle.anyshape_matrices(img, ps)
    g = torch.from_numpy(img).cuda().unsqueeze(0)
    ok = all(np.array_equal(ctx.planes_any(g, ps, c)[0].cpu().numpy().view(np.uint32), np.ascontiguousarray(want[c]).view(np.uint32)) for c in range(3))
    ranks = [int(rng.integers(1, 9)) for _ in range(3)]
    fac = [(rng.integers(-16, 16, (dims[c][4], ranks[c]), dtype=np.int8), rng.integers(-16, 16, (dims[c][5], ranks[c]), dtype=np.int8)) for c in range(3)]
    dec = ctx.decode_any([torch.from_numpy(f[0][None]).cuda() for f in fac], [torch.from_numpy(f[1][None]).cuda() for f in fac], H, W, ps)[0].cpu().numpy()
    ok2 = np.array_equal(dec, oracle.qmf_anyshape_decode(fac, H, W, ps))
    if not (ok and ok2):
        bad2 += 1
        print(f"GEOMETRY MISMATCH {H}x{W} patch {ps}: planes {ok} decode {ok2}", flush=True)
print(f"planes/decode: {trials} trials, {bad2} mismatching")
