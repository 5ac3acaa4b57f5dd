"""Timing of the any-shape branches (qmf_encode with patch sizes other than 8x8, or patch=False): B x 512x768 images at
one quality, per plane class: matrices, SVD initialisation, 10 BCD iterations.  Prints Mpixel/s of the whole
factorisation (inputs resident in HBM) and the per-kernel-class times.  Development aid; the contract benchmark of the
repository is bench.py."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
from lrf_amd.codec import anyshape_ranks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Q = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
H, W = 512, 768
g = torch.Generator(device="cuda").manual_seed(0)
base = torch.rand(B, 3, H // 8, W // 8, device="cuda", generator=g) * 255
imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear") + torch.randn(B, 3, H, W, device="cuda", generator=g) * 4
        ).clamp(0, 255).to(torch.uint8)
ctx = _lib.context(0)


def encode(ps, ranks):
    out = []
    for c in range(3):
        X = ctx.planes_any(imgs, ps, c)
        out.append(ctx.decompose(X, ranks[c], 10, -16, 15))
    return out


ONLY = sys.argv[3] if len(sys.argv) > 3 else None  # "4", "16", "32" or "none": one patch size only
for ps in ((4, 4), (16, 16), (32, 32), None):
    if ONLY is not None and ONLY != ("none" if ps is None else str(ps[0])):
        continue
    ranks = anyshape_ranks((H, W), ps, None, Q)
    dims = _lib.plane_dims_any(H, W, ps)
    encode(ps, ranks)
    torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    encode(ps, ranks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    k = {nm: round(ctx.kernel_time(i)[0], 2) for i, nm in _lib.KERNEL_NAMES.items() if ctx.kernel_time(i)[1]}
    print(f"patch {ps} quality {Q}: matrices {[(d[4], d[5]) for d in dims]} ranks {ranks}: {dt*1e3:.1f} ms per {B} images, "
          f"{B*H*W/dt/1e6:.0f} Mpix/s; ms by class {k}", flush=True)
