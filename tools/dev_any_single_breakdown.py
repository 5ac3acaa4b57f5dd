"""Where one patch=False encode of one image spends its wall time (development aid)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lrf_amd
from lrf_amd import _lib, codec
from conftest import config3_image
img = config3_image(3)
for _ in range(3): lrf_amd.qmf_encode(img, quality=20, patch=False)
dev = img.unsqueeze(0).cuda()
ctx = _lib.context(0)
ranks = codec.anyshape_ranks((512, 768), None, None, 20.0)
torch.cuda.synchronize()
for c in range(3):
    X = ctx.planes_any(dev, None, c)
    for _ in range(2): ctx.decompose(X, ranks[c], 10, -16, 15)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u, v = ctx.decompose(X, ranks[c], 10, -16, 15)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    u0, v0 = ctx.svd_init(X, ranks[c]); torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"plane {c} {tuple(X.shape)} R={ranks[c]}: decompose issue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms; svd_init alone {1e3*(t3-t2):.2f} ms", flush=True)
t0 = time.perf_counter(); s = lrf_amd.qmf_encode(img, quality=20, patch=False); t1 = time.perf_counter()
print(f"whole encode {1e3*(t1-t0):.2f} ms")
