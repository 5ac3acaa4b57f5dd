"""Pins the oracle's BCD (oracle/lrf_oracle.c lrf_oracle_bcd) against the REFERENCE on every matrix shape and rank the
patch-size / patch=False branches produce, and on ranks 8..64 of the 8x8 branch: from the reference's own SVD start, K
iterations of the reference's QMF(factor=(0,1)) against the oracle's, bit for bit (fp32 factors compared as values).
Build container only (imports the reference through tools/ref_loader.py, one thread).  Exits non-zero on a mismatch.
usage: PYTHONDONTWRITEBYTECODE=1 python tools/pin_oracle_anyshape.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np, torch
import ref_loader
from oracle import oracle
torch.set_num_threads(1)
ns = ref_loader.load()
g = torch.Generator().manual_seed(3)
bad = 0


def run(x, R, tag, K=10, bounds=(-16, 15)):
    global bad
    x = x.float()
    u0, v0, _ = ns.fqmf.SVDInit(rank=R)(x.unsqueeze(0))
    u, v, _ = ns.fqmf.QMF(rank=R, bounds=bounds, num_iters=K, factor=(0, 1)).decompose(x.unsqueeze(0))
    U, V = oracle.bcd(x.numpy(), u0[0].numpy(), v0[0].numpy(), K, bounds)
    du, dv = int((U != u[0].numpy()).sum()), int((V != v[0].numpy()).sum())
    bad += (du + dv) > 0
    print(f"{tag}: X {tuple(x.shape)} R={R} K={K} bounds={bounds}: U mismatches {du}, V mismatches {dv}", flush=True)


for (H, W) in [(512, 768), (173, 264)]:
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
    plane = ns.cutils.rgb_to_ycbcr(img.float())[0:1]
    for p in (4, 8, 16, 32):
        x = ns.cqmf.patchify(ns.cutils.pad_image(plane, (p, p), mode="reflect"), (p, p))
        for R in (1, 3, 8, 12, 20, 40, 64, 102, 154):
            if R <= min(x.shape):
                run(x, R, f"{H}x{W} patch {p}")
    for R in (1, 5, 20, 70, 130, 205):
        if R <= min(plane.shape[-2:]):
            run(plane[0], R, f"{H}x{W} no patches")
    run(plane[0], 33, f"{H}x{W} no patches", 10, (-128, 127))
    x4 = ns.cqmf.patchify(ns.cutils.pad_image(plane, (4, 4), mode="reflect"), (4, 4))
    run(x4, 16, f"{H}x{W} patch 4 full rank")
    run(x4, 5, f"{H}x{W} patch 4", 2, (-128, 127))
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
