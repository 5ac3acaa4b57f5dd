"""k_bcd_p<F16, NP32> (LRF_PERSIST=1: from 1024 blocks) against the launch-per-iteration kernels (chunks of eight images) and the
oracle, per plane, for the rank triples of argv (default: a few of the families 9..32) — where do they differ?  Development aid."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lrf_amd
from lrf_amd.codec import split_factors
from oracle import oracle
assert os.environ.get("LRF_PERSIST") == "1"
oracle.build()
H, W, B = 512, 768, 48
K = int(os.environ.get("K", "3"))
g = torch.Generator().manual_seed(17)
base = torch.rand(B, 3, H // 8, W // 8, generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
        + torch.randn(B, 3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8).cuda()
triples = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(23, 23, 23), (24, 24, 24), (20, 20, 20), (17, 17, 17), (23, 8, 8), (23, 12, 12)]
for ranks in triples:
    U, V = lrf_amd.qmf_factorize_batch(imgs, ranks, num_iters=K)
    Us, Vs = lrf_amd.qmf_factorize_batch(imgs[:8].clone(), ranks, num_iters=K)
    torch.cuda.synchronize()
    for b in (0, 7):
        X = oracle.rgb_to_planes(imgs[b].cpu().numpy())
        gp = split_factors(U[b].cpu().numpy(), V[b].cpu().numpy(), (H, W), ranks)
        gc = split_factors(Us[b].cpu().numpy(), Vs[b].cpu().numpy(), (H, W), ranks)
        msg = []
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, (-16, 15))
            u, v = u.astype(np.int8), v.astype(np.int8)
            du_p, dv_p = int((gp[2 * c] != u).sum()), int((gp[2 * c + 1] != v).sum())
            du_c, dv_c = int((gc[2 * c] != u).sum()), int((gc[2 * c + 1] != v).sum())
            rows_p = np.nonzero((gp[2 * c] != u).any(axis=1))[0]
            msg.append(f"plane {c}: persist dU {du_p} dV {dv_p} (rows {rows_p[:4]}..{rows_p[-2:] if len(rows_p) else ''} of {len(rows_p)}) | chunk dU {du_c} dV {dv_c}")
        print(ranks, "image", b, "; ".join(msg), flush=True)
