"""Child process of tests/test_configs_at_size.py::test_persistent_kernel_other_iteration_counts_bounds_and_ranks: started with
LRF_PERSIST=1 (the switch is read once per process), so calls of 1024 blocks or more iterate on k_bcd_p; calls of eight images
stay on the launch-per-iteration kernels.  Prints one "ok" line per case."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lrf_amd
from lrf_amd.codec import split_factors
from oracle import oracle

assert os.environ.get("LRF_PERSIST") == "1"
oracle.build()
for (H, W, B), cases in (((512, 768, 48), (((8, 1, 5), (-16, 15), 2), ((7, 3, 3), (-3, 5), 3), ((1, 2, 8), (-128, 127), 5), ((4, 4, 4), (-16, 15), 10))),
                         ((173, 264, 272), (((5, 8, 1), (-16, 15), 4), ((8, 8, 8), (-22, 22), 2)))):
    g = torch.Generator().manual_seed(17)
    base = torch.rand(B, 3, H // 8, W // 8, generator=g) * 255
    imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
            + torch.randn(B, 3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8).cuda()
    for ranks, bounds, K in cases:
        U, V = lrf_amd.qmf_factorize_batch(imgs, ranks, num_iters=K, bounds=bounds)
        for b0 in (0, B - 8):
            Us, Vs = lrf_amd.qmf_factorize_batch(imgs[b0:b0 + 8].clone(), ranks, num_iters=K, bounds=bounds)
            assert torch.equal(U[b0:b0 + 8], Us) and torch.equal(V[b0:b0 + 8], Vs), (ranks, bounds, K, b0)
        X = oracle.rgb_to_planes(imgs[B - 1].cpu().numpy())
        got = split_factors(U[B - 1].cpu().numpy(), V[B - 1].cpu().numpy(), (H, W), ranks)
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, bounds)
            assert np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8)), (ranks, bounds, K, c)
        print("ok", (H, W, B), ranks, bounds, K, flush=True)
ctx = lrf_amd._lib.context(0)
ctx.synchronize()  # raises if a poll of k_bcd_p expired
# and the launches above did go through k_bcd_p: one more call with that kernel's timer on
ctx.profile_kernels([lrf_amd._lib.LRF_K_BCD, lrf_amd._lib.LRF_K_BCD_PERSIST])
ctx.profile_reset()
lrf_amd.qmf_factorize_batch(imgs, (8, 8, 8), num_iters=2, bounds=(-22, 22))
torch.cuda.synchronize()
assert ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST)[1] == 1, ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST)
ctx.profile(False)
print("persistent launches seen", flush=True)
