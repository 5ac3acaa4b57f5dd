"""Edge cases of the any-shape branches on the GPU (development aid): zero / constant planes, tiny and thin images.
Prints stream length and MSE; the reference's values (run in the build container, one thread) are listed beside them."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
cases = [("zero nopatch", torch.zeros(3, 24, 40, dtype=torch.uint8), dict(quality=20, patch=False), (491, 4.3333)),
         ("const p16", torch.full((3, 40, 56), 77, dtype=torch.uint8), dict(quality=10, patch_size=(16, 16)), (630, 4.0)),
         ("const nopatch", torch.full((3, 24, 40), 200, dtype=torch.uint8), dict(quality=30, patch=False), (510, 0.0)),
         ("tiny p4", torch.arange(3 * 8 * 8, dtype=torch.uint8).reshape(3, 8, 8), dict(quality=50, patch_size=(4, 4)), (660, 36.4375)),
         ("thin nopatch", (torch.arange(3 * 9 * 200) % 251).to(torch.uint8).reshape(3, 9, 200), dict(quality=40, patch=False), (664, 3847.12))]
for name, img, kw, ref in cases:
    enc = lrf_amd.qmf_encode(img, **kw)
    dec = lrf_amd.qmf_decode(enc)
    mse = float(((img.float() - dec.float()) ** 2).mean())
    print(f"{name}: {len(enc)} bytes, mse {mse:.4f}   (reference: {ref[0]} bytes, mse {ref[1]})")
