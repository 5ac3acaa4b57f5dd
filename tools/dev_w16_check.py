"""Diagnostic: k_bcd_w16 (one wave per block, ranks <= 16, iterations >= 2) against the workgroup kernel k_bcd<., 16> and the
oracle.  Large batches take the wave kernel (>= LRF_BCDW16_MIN_BLOCKS blocks), chunks of 8 images the workgroup kernel."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, lrf_amd
from lrf_amd.codec import split_factors
from oracle import oracle
oracle.build()
bad = 0
for (H, W, B) in ((173, 264, 272), (512, 768, 48), (64, 96, 1100)):
    g = torch.Generator().manual_seed(5)
    base = torch.rand(B, 3, max(H // 8, 2), max(W // 8, 2), generator=g) * 255
    imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
            + torch.randn(B, 3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8).cuda()
    for ranks in ((9, 1, 2), (10, 3, 4), (11, 5, 6), (12, 7, 8), (13, 14, 15), (16, 16, 9), (16, 8, 8), (10, 5, 5)):
        for bounds in ((-16, 15), (-3, 5)):
            U, V = lrf_amd.qmf_factorize_batch(imgs, ranks, num_iters=4, bounds=bounds)
            ok = True
            for b0 in (0, 8, B - 8):
                Us, Vs = lrf_amd.qmf_factorize_batch(imgs[b0:b0 + 8].clone(), ranks, num_iters=4, bounds=bounds)
                ok &= torch.equal(U[b0:b0 + 8], Us) and torch.equal(V[b0:b0 + 8], Vs)
            X = oracle.rgb_to_planes(imgs[B - 1].cpu().numpy())
            got = split_factors(U[B - 1].cpu().numpy(), V[B - 1].cpu().numpy(), (H, W), ranks)
            for c in range(3):
                u, v = oracle.qmf_decompose(X[c], ranks[c], 4, bounds)
                ok &= np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8))
            bad += not ok
            print(f"{H}x{W} B={B} ranks {ranks} bounds {bounds}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
