"""Development aid: 64-column matrices of ranks 17..64 through k_bcd_big (default) or through the any-shape kernels
(LRF_FORCE_ANY=1): luma [64, 6144, 64] and chroma [128, 1536, 64] matrices as one 64-image batch has them."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from lrf_amd import _lib
ctx = _lib.context(0)
XL = torch.rand(64, 6144, 64, device="cuda") * 255
XC = torch.rand(128, 1536, 64, device="cuda") * 255
for ranks in ((20, 10), (26, 13), (40, 20), (64, 32)):
    def fn():
        ctx.decompose(XL, ranks[0], 10, -16, 15); ctx.decompose(XC, ranks[1], 10, -16, 15)
    fn(); torch.cuda.synchronize(); ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ctx.profile(False)
    k = {nm: round(ctx.kernel_time(i)[0], 2) for i, nm in _lib.KERNEL_NAMES.items() if ctx.kernel_time(i)[1]}
    print(f"LRF_FORCE_ANY={os.environ.get('LRF_FORCE_ANY', '0')} ranks {ranks}: {dt*1e3:.2f} ms; by class {k}", flush=True)
