"""Randomised HIP-vs-oracle sweep aimed at k_bcd_w32 / k_bcd_w32f (ranks 17..32, bit for bit): random shapes (ragged blocks,
one to several blocks per plane), ranks 17..32 per plane (a third of the cases with one rank in every plane, which also puts
the first iteration on k_bcd_w32f), iteration counts, bounds inside and outside the int16-table range.  Run with
LRF_FAMILY_SPLIT_BLOCKS=1 LRF_BCDW32_MIN_BLOCKS=1 so that runs of any size take the wave kernels.  Exits non-zero on a mismatch."""
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
    H = rnd.choice([64, 99, 128, 173, 200, 256, 333, 512])
    W = rnd.choice([96, 130, 192, 264, 300, 384, 768])
    B = rnd.choice([1, 2, 3, 6])
    if rnd.random() < 0.35:
        r = rnd.randint(17, 32)
        ranks = (r, r, r)
    else:
        ranks = tuple(rnd.randint(17, 32) for _ in range(3))
    K = rnd.choice([1, 2, 3, 5])
    bounds = rnd.choice([(-16, 15), (-16, 15), (-8, 7), (-22, 22), (-3, 5), (0, 15), (-128, 127), (-32, 31)])
    kind = rnd.choice(["rand", "smooth", "smooth"])
    g = torch.Generator().manual_seed(5000 + i)
    if kind == "rand":
        img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    else:
        base = torch.rand(B, 3, max(H // 8, 1), max(W // 8, 1), generator=g) * 255
        img = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
               + torch.randn(B, 3, H, W, generator=g) * 5).clamp(0, 255).to(torch.uint8)
    U, V = lrf_amd.qmf_factorize_batch(img.cuda(), ranks, num_iters=K, bounds=bounds)
    ok = True
    for b in range(B):
        got = split_factors(U[b].cpu().numpy(), V[b].cpu().numpy(), (H, W), ranks)
        X = oracle.rgb_to_planes(img[b].numpy())
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, bounds)
            if not (np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8))):
                ok = False
                print(f"   MISMATCH image {b} plane {c}: U {int((got[2 * c] != u.astype(np.int8)).sum())} V {int((got[2 * c + 1] != v.astype(np.int8)).sum())} entries")
    print(f"[{i}] {H}x{W} B={B} ranks={ranks} K={K} bounds={bounds} {kind}: {'ok' if ok else 'FAIL'}", flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
