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
ctx = lrf_amd._lib.context(0)
# (rank triple, bounds, K, persistent launches expected): ranks <= 8 (round 4), then the families 9..16 and 17..32 and their
# mixes (round 5: k_bcd_p<F16, NP32>) — the reference's quality sweep (16,8,8) (20,10,10) (26,13,13), a luma rank whose chroma
# planes fall into the <= 8 family (17,8,8), the largest (32,16,16), one family for all planes, an odd pair count, K = 2;
# bounds outside the exact-integer range of a family ((-128,127) at rank 12; (-25,25): |b| beyond int16 at rank 20) must fall
# back to the launch-per-iteration kernels and still agree
LOW = (((8, 1, 5), (-16, 15), 2, 1), ((7, 3, 3), (-3, 5), 3, 1), ((1, 2, 8), (-128, 127), 5, 1), ((4, 4, 4), (-16, 15), 10, 1))
MIX = (((16, 8, 8), (-16, 15), 10, 1), ((10, 5, 5), (-16, 15), 3, 1), ((12, 12, 12), (-8, 7), 4, 1), ((20, 10, 10), (-16, 15), 5, 1),
       ((26, 13, 13), (-16, 15), 10, 1), ((17, 8, 8), (-22, 22), 2, 1), ((32, 16, 16), (-16, 15), 3, 1), ((23, 23, 23), (-16, 15), 3, 1),
       ((12, 6, 6), (-128, 127), 3, 0), ((20, 10, 10), (-25, 25), 3, 0),
       # every pair count NP = 9..16 of ranks 17..32 is an instantiation of its own (k_bcd_p<true, NP>; <true, 12> alone was
       # miscompiled until its operand loop became a compile-time expansion): 19 / 22 / 25 / 28 / 30 with NP = 10, 11, 13, 14, 15
       ((19, 9, 9), (-16, 15), 3, 1), ((22, 11, 11), (-16, 15), 3, 1), ((25, 12, 12), (-16, 15), 3, 1), ((28, 14, 14), (-16, 15), 3, 1),
       ((30, 15, 15), (-16, 15), 3, 1), ((24, 12, 6), (-16, 15), 3, 1))
for (H, W, B), cases in (((512, 768, 48), LOW + MIX),
                         ((173, 264, 272), (((5, 8, 1), (-16, 15), 4, 1), ((8, 8, 8), (-22, 22), 2, 1), ((13, 6, 9), (-16, 15), 4, 1),
                                            ((18, 9, 4), (-16, 15), 3, 1)))):
    g = torch.Generator().manual_seed(17)
    base = torch.rand(B, 3, H // 8, W // 8, generator=g) * 255
    imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
            + torch.randn(B, 3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8).cuda()
    for ranks, bounds, K, want_persist in cases:
        ctx.profile_kernels([lrf_amd._lib.LRF_K_BCD, lrf_amd._lib.LRF_K_BCD_PERSIST])
        ctx.profile_reset()
        U, V = lrf_amd.qmf_factorize_batch(imgs, ranks, num_iters=K, bounds=bounds)
        torch.cuda.synchronize()
        assert ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST)[1] == want_persist, (ranks, bounds, K, ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST))
        ctx.profile(False)
        for b0 in (0, B - 8):
            Us, Vs = lrf_amd.qmf_factorize_batch(imgs[b0:b0 + 8].clone(), ranks, num_iters=K, bounds=bounds)
            assert torch.equal(U[b0:b0 + 8], Us) and torch.equal(V[b0:b0 + 8], Vs), (ranks, bounds, K, b0)
        X = oracle.rgb_to_planes(imgs[B - 1].cpu().numpy())
        got = split_factors(U[B - 1].cpu().numpy(), V[B - 1].cpu().numpy(), (H, W), ranks)
        for c in range(3):
            u, v = oracle.qmf_decompose(X[c], ranks[c], K, bounds)
            assert np.array_equal(got[2 * c], u.astype(np.int8)) and np.array_equal(got[2 * c + 1], v.astype(np.int8)), (ranks, bounds, K, c)
        print("ok", (H, W, B), ranks, bounds, K, flush=True)
ctx.synchronize()  # raises if a poll of k_bcd_p expired
# and the launches above did go through k_bcd_p: one more call with that kernel's timer on
ctx.profile_kernels([lrf_amd._lib.LRF_K_BCD, lrf_amd._lib.LRF_K_BCD_PERSIST])
ctx.profile_reset()
lrf_amd.qmf_factorize_batch(imgs, (8, 8, 8), num_iters=2, bounds=(-22, 22))
torch.cuda.synchronize()
assert ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST)[1] == 1, ctx.kernel_time(lrf_amd._lib.LRF_K_BCD_PERSIST)
ctx.profile(False)
print("persistent launches seen", flush=True)
