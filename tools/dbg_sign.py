import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import lrf_amd
from lrf_amd import _lib
from lrf_amd.codec import split_factors
from conftest import Case
from oracle import oracle
for name in ("tiny_q7", "tiny_r7", "tiny_rank2"):
    case = Case(name)
    H, W = case.image.shape[-2:]
    sign = np.concatenate(case.signs())
    print(name, case.ranks, "signs", sign.tolist())
    U, V = lrf_amd.qmf_factorize_batch(case.image.cuda().unsqueeze(0), case.ranks, init_sign=sign)
    got = split_factors(U[0].cpu().numpy(), V[0].cpu().numpy(), (H, W), case.ranks)
    X = oracle.rgb_to_planes(case.image.numpy())
    ref = case.ref_factors()
    ctx = _lib.context(0)
    for c in range(3):
        u, v = oracle.qmf_decompose(X[c], case.ranks[c], 10, (-16, 15), sign=case.z[f"sign{c}"])
        print(f"  plane {c}: HIP vs oracle U {int((got[2*c]!=u).sum())} V {int((got[2*c+1]!=v).sum())} | oracle vs ref U {int((u!=ref[2*c]).sum())} V {int((v!=ref[2*c+1]).sum())}")
        sg = torch.from_numpy(case.z[f"sign{c}"]).cuda().reshape(1, -1)
        u0, v0 = ctx.svd_init(torch.from_numpy(X[c]).cuda().unsqueeze(0), case.ranks[c], sg)
        ou, ov = oracle.svd_init(X[c], case.ranks[c], sign=case.z[f"sign{c}"])
        print(f"     init with sign: v0 mism {int((v0[0].cpu().numpy()!=ov).sum())} u0 mism {int((u0[0].cpu().numpy()!=ou).sum())}")
