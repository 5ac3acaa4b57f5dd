"""Developer check, stage by stage, HIP vs oracle (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lrf_amd  # noqa: E402
from lrf_amd import _lib  # noqa: E402
from lrf_amd.codec import split_factors  # noqa: E402
from oracle import oracle  # noqa: E402


def neq(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return int((a.view(np.uint8) != b.view(np.uint8)).reshape(a.shape + (-1,)).any(-1).sum())


def main():
    ctx = _lib.context(0)
    for (H, W, seed) in [(64, 96, 1), (173, 264, 2), (512, 768, 0)]:
        g = torch.Generator().manual_seed(seed)
        img = torch.randint(0, 256, (2, 3, H, W), dtype=torch.uint8, generator=g)
        dims = _lib.plane_dims(H, W)
        X = ctx.planes_from_rgb(img.cuda())
        torch.cuda.synchronize()
        Xh = X.cpu().numpy()
        for b in range(2):
            Xo = oracle.rgb_to_planes(img[b].numpy())
            off = 0
            for c in range(3):
                M = dims[c][4]
                got = Xh[b, off:off + M * 64].reshape(M, 64)
                off += M * 64
                print(f"{H}x{W} img{b} plane{c} X mismatches: {neq(got, Xo[c])}/{got.size}")
        for ranks in ((4, 2, 2), (7, 3, 3)):
            # init
            Xo = oracle.rgb_to_planes(img[0].numpy())
            for c in range(3):
                M, R = dims[c][4], ranks[c]
                xd = torch.from_numpy(Xo[c]).cuda().unsqueeze(0)
                u0, v0 = ctx.svd_init(xd, R)
                torch.cuda.synchronize()
                ou0, ov0 = oracle.svd_init(Xo[c], R)
                print(f"  init plane{c} R{R}: v0 mism {neq(v0[0].cpu().numpy(), ov0)}/{ov0.size} maxdiff "
                      f"{np.abs(v0[0].cpu().numpy() - ov0).max():.2e}  u0 mism {neq(u0[0].cpu().numpy(), ou0)}/{ou0.size} "
                      f"maxdiff {np.abs(u0[0].cpu().numpy() - ou0).max():.2e}")
                # bcd from oracle init
                for K in (1, 2, 10):
                    U, V = ctx.bcd(xd, torch.from_numpy(ou0).cuda().unsqueeze(0), torch.from_numpy(ov0).cuda().unsqueeze(0), K, -16, 15)
                    torch.cuda.synchronize()
                    uo, vo = oracle.bcd(Xo[c], ou0, ov0, K)
                    print(f"    bcd K={K}: U mism {neq(U[0].cpu().numpy(), uo.astype(np.int8))}/{uo.size} V mism "
                          f"{neq(V[0].cpu().numpy(), vo.astype(np.int8))}/{vo.size}")
            t = time.time()
            U, V = lrf_amd.qmf_factorize_batch(img.cuda(), ranks)
            torch.cuda.synchronize()
            dt = time.time() - t
            for b in range(2):
                got = split_factors(U[b].cpu().numpy(), V[b].cpu().numpy(), (H, W), ranks)
                Xo = oracle.rgb_to_planes(img[b].numpy())
                tot = 0
                for c in range(3):
                    uo, vo = oracle.qmf_decompose(Xo[c], ranks[c], 10)
                    tot += neq(got[2 * c], uo.astype(np.int8)) + neq(got[2 * c + 1], vo.astype(np.int8))
                dec = ctx.decode_rgb(U[b:b + 1], V[b:b + 1], H, W, ranks)[0].cpu().numpy()
                ref = oracle.planes_to_rgb(got[0::2], got[1::2], H, W)
                print(f"  fused encode ranks {ranks} img{b}: factor mismatches {tot}; decode mismatches {neq(dec, ref)}/{dec.size}"
                      f"  ({dt*1e3:.1f} ms)")


if __name__ == "__main__":
    main()
