"""Randomised HIP-vs-oracle parity sweep (bit for bit): shapes, ranks, iteration counts, bounds, batch sizes.
Run on the GPU box; prints one line per configuration and exits non-zero on the first mismatch."""
import os, sys, random
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, lrf_amd
from lrf_amd.codec import split_factors
from oracle import oracle
oracle.build()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for i in range(n_cfg):
    H = rnd.choice([16, 24, 40, 64, 99, 128, 173, 200, 256, 333, 512])
    W = rnd.choice([16, 32, 56, 96, 130, 192, 264, 300, 384, 768])
    B = rnd.choice([1, 1, 2, 3])
    top = rnd.choice([8, 8, 8, 16, 16, 24])
    ranks = tuple(rnd.randint(1, top) for _ in range(3))
    K = rnd.choice([1, 2, 3, 5, 10])
    bounds = rnd.choice([(-16, 15), (-16, 15), (-8, 7), (-128, 127), (-1, 1), (0, 15)])
    kind = rnd.choice(["rand", "smooth", "const"])
    g = torch.Generator().manual_seed(1000 + i)
    if kind == "rand":
        img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    elif kind == "smooth":
        base = torch.rand(B, 3, max(H // 8, 1), max(W // 8, 1), generator=g) * 255
        img = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
               + torch.randn(B, 3, H, W, generator=g) * 3).clamp(0, 255).to(torch.uint8)
    else:
        img = torch.full((B, 3, H, W), int(rnd.randint(0, 255)), dtype=torch.uint8)
    dims = lrf_amd._lib.plane_dims(H, W)
    ranks = tuple(min(r, 64) for r in ranks)
    try:
        U, V = lrf_amd.qmf_factorize_batch(img.cuda(), ranks, num_iters=K, bounds=bounds)
    except (NotImplementedError, ValueError) as e:
        print(f"[{i}] {H}x{W} B={B} ranks={ranks} K={K} bounds={bounds} {kind}: rejected ({str(e)[:60]})")
        continue
    ok = True
    for b in range(B):
        got = split_factors(U[b].cpu().numpy(), V[b].cpu().numpy(), (H, W), ranks)
        X = oracle.rgb_to_planes(img[b].numpy())
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, bounds)
            if not (np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8))):
                ok = False
                du = int((got[2 * c] != u.astype(np.int8)).sum()); dv = int((got[2 * c + 1] != v.astype(np.int8)).sum())
                print(f"   MISMATCH image {b} plane {c}: U {du} V {dv} entries")
    print(f"[{i}] {H}x{W} B={B} ranks={ranks} K={K} bounds={bounds} {kind}: {'ok' if ok else 'FAIL'}")
    bad += not ok
sys.exit(1 if bad else 0)
