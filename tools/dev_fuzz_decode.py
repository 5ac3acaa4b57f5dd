"""Randomised HIP-vs-oracle parity sweep of the decoder (bit for bit): random int8 factors, shapes, ranks."""
import os, sys, random
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, lrf_amd
from lrf_amd import _lib
from oracle import oracle
oracle.build()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = _lib.context(0)
bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H = rnd.choice([8, 9, 16, 23, 40, 64, 99, 173, 256, 333, 512]); W = rnd.choice([8, 11, 16, 32, 57, 96, 130, 264, 384, 768])
    top = rnd.choice([4, 8, 8, 16, 40])
    if len(sys.argv) > 3 and sys.argv[3] == "strip":  # the sizes k_decode_strip takes: any height, four-aligned chroma columns, ranks <= 8
        H = rnd.choice([8, 9, 15, 17, 23, 25, 31, 33, 41, 99, 173, 255, 333, 511, 683]); W = rnd.choice([16, 32, 48, 60, 124, 252, 508, 1020, 96, 2048, 1040])
        top = rnd.choice([4, 8, 8, 16, 32, 40])
    ranks = tuple(rnd.randint(1, top) for _ in range(3)); amp = rnd.choice([3, 16, 127])
   
    try:
        dims = _lib.plane_dims(H, W)
    except ValueError as e:
        print(f"[{i}] {H}x{W}: rejected ({str(e)[:50]})"); continue
    rng = np.random.default_rng(i)
    Us = [rng.integers(-amp, amp + 1, (d[4], r), dtype=np.int8) for d, r in zip(dims, ranks)]
    Vs = [rng.integers(-amp, amp + 1, (64, r), dtype=np.int8) for r in ranks]
    U = torch.from_numpy(np.concatenate([u.ravel() for u in Us]))[None].cuda()
    V = torch.from_numpy(np.concatenate([v.ravel() for v in Vs]))[None].cuda()
    try:
        got = ctx.decode_rgb(U, V, H, W, ranks)[0].cpu().numpy()
    except (ValueError, NotImplementedError) as e:
        print(f"[{i}] {H}x{W} ranks={ranks}: rejected ({str(e)[:50]})"); continue
    want = oracle.planes_to_rgb(Us, Vs, H, W)
    ok = np.array_equal(got, want)
    print(f"[{i}] {H}x{W} ranks={ranks} amp={amp}: {'ok' if ok else 'FAIL %d px' % int((got != want).sum())}")
    bad += not ok
sys.exit(1 if bad else 0)
