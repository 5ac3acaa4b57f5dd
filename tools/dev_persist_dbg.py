"""Debug aid for k_bcd_p: factors of a persistent run (this process: LRF_PERSIST=1) against the oracle for K = 2, 3 on a few images."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
import numpy as np, torch, lrf_amd
from lrf_amd.codec import split_factors
from oracle import oracle
g = torch.Generator().manual_seed(3)
B, H, W, ranks = 48, 512, 768, (7, 3, 3)
imgs = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
for K in (2, 3, 4):
    U, V = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, num_iters=K)
    torch.cuda.synchronize()
    for b in range(B):
        got = split_factors(U[b].cpu().numpy(), V[b].cpu().numpy(), (H, W), ranks)
        X = oracle.rgb_to_planes(imgs[b].numpy())
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, (-16, 15))
            du = (got[2 * c] != u.astype(np.int8)); dv = (got[2 * c + 1] != v.astype(np.int8))
            rows = np.nonzero(du.any(axis=1))[0]
            if du.sum() or dv.sum(): print(f"K={K} image {b} plane {c}: U diff {int(du.sum())} entries in {len(rows)} rows (first rows {rows[:6].tolist()}, blocks {sorted(set((rows // 384).tolist()))[:8]}), V diff {int(dv.sum())}")
K = 2
U, V = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, num_iters=K)
U2, V2 = lrf_amd.qmf_factorize_batch(imgs.cuda(), ranks, num_iters=K)
print("two runs equal:", torch.equal(U, U2), torch.equal(V, V2))
got = split_factors(U[0].cpu().numpy(), V[0].cpu().numpy(), (H, W), ranks)
X = oracle.rgb_to_planes(imgs[0].numpy())
u, v = oracle.qmf_decompose(X[0], 7, K, (-16, 15))
u1, v1 = oracle.qmf_decompose(X[0], 7, 1, (-16, 15))
for r in range(0, 12):
    print(r, "got", got[0][r].tolist(), "want", u[r].astype(int).tolist(), "iter1", u1[r].astype(int).tolist())
flat_g = got[0].reshape(-1)[:120]; flat_w = u.astype(np.int8).reshape(-1)[:120]
print("first differing byte offsets:", np.nonzero(flat_g != flat_w)[0][:40].tolist())
wrong = np.nonzero((got[0][:384] != u.astype(np.int8)[:384]).any(axis=1))[0]
right = sorted(set(range(384)) - set(wrong.tolist()))
print("right rows of block 0:", right)
print("wrong rows by sub-tile:", [int(((wrong >= 64 * t) & (wrong < 64 * t + 64)).sum()) for t in range(6)])
# is the stored row what one gets from the NEW u of the same sub-tile as old input?  compare columns
