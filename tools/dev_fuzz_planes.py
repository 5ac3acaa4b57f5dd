"""Diagnostic: the patch-matrix kernels (k_planes16 / k_planes_strip / k_planes) against the oracle on random sizes —
odd and even sides, sides below 16, widths that leave the last column group partly empty, batches of 1..3 —
bit for bit; then the time of the CLIC-sized batch (512 x 1365x2048)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
from oracle import oracle
oracle.build()
ctx = _lib.context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sizes = [(8, 8), (9, 9), (15, 17), (16, 16), (17, 31), (24, 40), (173, 264), (255, 257), (1365, 2048), (33, 1030), (662, 992), (40, 8), (8, 300)]
sizes += [(int(rng.integers(8, 400)), int(rng.integers(8, 600))) for _ in range(40)]
bad = 0
for (H, W) in sizes:
    B = int(rng.integers(1, 4)) if H * W < 300000 else 1
    img = torch.from_numpy(rng.integers(0, 256, (B, 3, H, W), dtype=np.uint8))
    X = ctx.planes_from_rgb(img.cuda()).cpu().numpy()
    ok = True
    for b in range(B):
        want = oracle.rgb_to_planes(img[b].numpy())
        off = 0
        for c, d in enumerate(_lib.plane_dims(H, W)):
            got = X[b, off:off + d[4] * 64].reshape(d[4], 64)
            off += d[4] * 64
            if not np.array_equal(got.view(np.int32), want[c].view(np.int32)):
                ok = False
                idx = np.argwhere(got.view(np.int32) != want[c].view(np.int32))
                print(f"   {H}x{W} image {b} plane {c}: {len(idx)} differing entries, first at {idx[0]}")
    bad += not ok
    print(f"{H}x{W} B={B}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatching sizes:", bad)
if bad:
    sys.exit(1)
g = torch.Generator(device="cuda").manual_seed(1)
imgs = torch.randint(0, 256, (512, 3, 1365, 2048), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(3):
    X = ctx.planes_from_rgb(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    X = ctx.planes_from_rgb(imgs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
alg = imgs.numel() + X.numel() * 4
print(f"512 x 1365x2048: {dt*1e3:.3f} ms per batch (incl. the output allocation), {alg/dt/1e12:.2f} TB/s algorithmic")
