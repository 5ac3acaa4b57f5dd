"""Development aid: the RGB colour-space entry point (patchify + MFMA Gram + any-shape factorisation) beside the plain any-shape
call on matrices of the same shape, 64 x 512x768 images, rank argv[1] (default 10), K = 10.  When this tool was written the
first line ran a dedicated [M,192] kernel set (k_bcdn, since removed): 19.1 ms against 11.9 ms decided the switch."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
B, R = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
ctx = _lib.context(0)
X = torch.rand(B, 6144, 192, device="cuda") * 255
for name, fn in (("rgbspace entry point", lambda: ctx.qmf_rgbspace_encode(imgs, R)), ("any-shape path", lambda: ctx.decompose(X, R, 10, -16, 15))):
    fn(); torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ctx.profile(False)
    k = {nm: round(ctx.kernel_time(i)[0], 2) for i, nm in _lib.KERNEL_NAMES.items() if ctx.kernel_time(i)[1]}
    print(f"{name}: {dt*1e3:.1f} ms per {B} images; by class {k}", flush=True)
