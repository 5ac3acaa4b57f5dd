"""Child process of tests/test_fused_gram.py: prints one line per case with a digest of the int8 factors of an encode call.  The
parent runs it with LRF_FUSED_GRAM_MIN_CHUNKS=1 (k_planes16_gram for every call whose sides are multiples of 16) and with the
variable unset (the two-kernel form at these sizes) and compares the lines: the variable is read once per process."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lrf_amd
from lrf_amd import _lib

ctx = _lib.context(0)
g = torch.Generator().manual_seed(23)


def batch(B, H, W):
    base = torch.rand(B, 3, H // 8 + 1, W // 8 + 1, generator=g) * 255
    return (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
            + torch.randn(B, 3, H, W, generator=g) * 6).clamp(0, 255).to(torch.uint8)


def digest(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


# sizes: one unit per strip and a ragged last unit (W / 8 = 12, 34, 98 patch columns), one strip, the bench size; batch sizes
# whose chunk count is below / above a round of workgroups; all three rank families and a mix
for (B, H, W), ranks in (((1, 16, 16), (1, 1, 1)), ((3, 64, 96), (7, 3, 3)), ((5, 48, 272), (4, 2, 2)), ((2, 784, 16), (3, 3, 3)),
                         ((9, 256, 784), (12, 6, 6)), ((40, 512, 768), (7, 3, 3)), ((40, 512, 768), (20, 10, 10)),
                         ((160, 512, 768), (16, 8, 8))):
    imgs = batch(B, H, W).cuda()
    U, V = lrf_amd.qmf_factorize_batch(imgs, ranks)
    print("encode", (B, H, W), ranks, digest(U, V), flush=True)
# the sweep entry point (one luma plane per image computes an initialisation, the others share it) and the host -> host pipe
imgs = batch(6, 96, 160)
out = ctx.encode_sweep_rgb(imgs.cuda(), [(3, 1, 2), (12, 6, 6), (21, 10, 10), (7, 3, 3)], 5, -16, 15)
print("sweep", digest(*[t for pair in out for t in pair]), flush=True)
host = batch(48, 128, 192).pin_memory()
Uh, Vh = lrf_amd.qmf_factorize_host(host, (7, 3, 3))
print("pipe", digest(Uh, Vh), flush=True)
# which kernel formed the patch matrices (not part of the comparison: the parent checks it per run)
ctx.profile(True); ctx.profile_reset()
lrf_amd.qmf_factorize_batch(batch(3, 64, 96).cuda(), (7, 3, 3))
torch.cuda.synchronize()
print("launches: k_planes16_gram", ctx.kernel_time(_lib.LRF_K_PLANES_GRAM)[1], "k_planes16", ctx.kernel_time(_lib.LRF_K_PLANES)[1], file=sys.stderr, flush=True)
ctx.profile(False)
