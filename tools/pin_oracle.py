"""Pins oracle/lrf_oracle.c against the reference (runs only where /root/reference exists).

Stage-by-stage bit comparison on several images; prints mismatch counts.  The reference runs with
torch.set_num_threads(1) (its long X^T U reduction is thread-count dependent, see lrf_oracle.c).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ref_loader  # noqa: E402
from oracle import oracle  # noqa: E402


def test_images():
    g = torch.Generator().manual_seed(0)
    imgs = {"rand_512x768": torch.randint(0, 256, (3, 512, 768), dtype=torch.uint8, generator=g)}
    g = torch.Generator().manual_seed(1)
    imgs["rand_173x264"] = torch.randint(0, 256, (3, 173, 264), dtype=torch.uint8, generator=g)
    imgs["rand_64x96"] = torch.randint(0, 256, (3, 64, 96), dtype=torch.uint8, generator=g)
    # smooth synthetic: bilinear-upsampled noise + small noise
    g = torch.Generator().manual_seed(2)
    base = torch.rand(1, 3, 32, 48, generator=g) * 255
    sm = torch.nn.functional.interpolate(base, size=(256, 384), mode="bilinear", align_corners=False)[0]
    sm = (sm + torch.randn(sm.shape, generator=g) * 4).clamp(0, 255).to(torch.uint8)
    imgs["smooth_256x384"] = sm
    png = os.path.join(ref_loader.REF_ROOT, "figures", "kodim01.png")
    try:
        from PIL import Image
        im = np.asarray(Image.open(png).convert("RGB"))
        imgs["kodim01_fig_662x992"] = torch.from_numpy(im.copy()).permute(2, 0, 1).contiguous()
    except Exception as e:  # PIL may be absent
        print("note: natural image not loaded:", e)
    return imgs


def ref_planes(ns, img):
    ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
    chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=(0.5, 0.5), mode="area")
    return [ns.cqmf.patchify(ns.cutils.pad_image(c, (8, 8), mode="reflect"), (8, 8)) for c in chans]


def neq(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return int((a.view(np.int32) != b.view(np.int32)).sum())


def main():
    torch.set_num_threads(1)
    ns = ref_loader.load()
    oracle.build()
    for name, img in test_images().items():
        print(f"=== {name} {tuple(img.shape)}")
        Xr = ref_planes(ns, img)
        Xo = oracle.rgb_to_planes(img.numpy())
        for c in range(3):
            print(f"  planes[{c}] {tuple(Xr[c].shape)} mismatches {neq(Xr[c].numpy(), Xo[c])}")
        for ranks in ((4, 2, 2), (7, 3, 3)):
            for c in range(3):
                x = Xr[c]
                R = ranks[c]
                qmf = ns.fqmf.QMF(rank=R, bounds=(-16, 15), factor=(0, 1))
                u, v, w = qmf.init(x.unsqueeze(0).float())
                u0, v0 = u[0].numpy().copy(), v[0].numpy().copy()
                res = []
                for it in range(1, 11):
                    u, v, w = qmf.solver(x.unsqueeze(0).float(), [u, v, w])
                    if it in (1, 2, 10):
                        uo, vo = oracle.bcd(x.numpy(), u0, v0, it)
                        res.append((it, neq(u[0].numpy(), uo), neq(v[0].numpy(), vo)))
                print(f"  BCD from ref init  R={R} plane {c}: " + "  ".join(f"it{it}: U {a} V {b}" for it, a, b in res))
        # decode parity: decode the reference's bytes with the reference, and its factors with the oracle
        enc = ns.cqmf.qmf_encode(img, quality=7)
        dec = ns.cqmf.qmf_decode(enc).numpy()
        meta, fac = ns.cutils.separate_bytes(enc, 2)
        facs = [ns.cutils.decode_tensor(f).numpy() for f in ns.cutils.separate_bytes(fac, 6)]
        H, W = img.shape[-2:]
        deco = oracle.planes_to_rgb(facs[0::2], facs[1::2], H, W)
        print(f"  decode mismatching bytes: {int((dec != deco).sum())} / {dec.size}")


if __name__ == "__main__":
    main()
