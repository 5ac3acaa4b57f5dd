"""Generates tests/golden/*.npz by running the REFERENCE (imported from /root/reference) here.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
The reference runs with torch.set_num_threads(1): its long X^T U reduction (MKL sgemm, K = M) is
summed in a thread-count-dependent order, and one thread is the reproducible one.
Fixtures hold data only: inputs (or the seed recipe + sha256 for large ones), parameters, the
reference's encoded bytes, decoded image (or its sha256), PSNR / bpp, the LAPACK column signs of the
reference's SVD initialisation, and for small planes the initial factors (u0, v0).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def make_image(spec):
    kind = spec["kind"]
    if kind == "randint":
        g = torch.Generator().manual_seed(spec["seed"])
        return torch.randint(0, 256, (3, spec["H"], spec["W"]), dtype=torch.uint8, generator=g)
    if kind == "smooth":
        g = torch.Generator().manual_seed(spec["seed"])
        H, W = spec["H"], spec["W"]
        base = torch.rand(1, 3, H // 8, W // 8, generator=g) * 255
        sm = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)[0]
        return (sm + torch.randn(sm.shape, generator=g) * 4).clamp(0, 255).to(torch.uint8)
    if kind == "const":
        return torch.full((3, spec["H"], spec["W"]), spec["value"], dtype=torch.uint8)
    if kind == "natural":
        from PIL import Image
        im = np.asarray(Image.open(os.path.join(ref_loader.REF_ROOT, "figures", "kodim01.png")).convert("RGB"))
        return torch.from_numpy(im.copy()).permute(2, 0, 1).contiguous()
    raise ValueError(kind)


def wsign(v):
    """sign of sum_j (j+1) v[j, r] per column — the convention the build's init uses."""
    w = np.arange(1, v.shape[0] + 1, dtype=np.float64)[:, None]
    s = np.sign((w * v.astype(np.float64)).sum(0))
    s[s == 0] = -1
    return s.astype(np.int8)


CASES = [
    # name, image spec, encoder kwargs, store_image, store_init
    ("tiny_q7", dict(kind="randint", seed=11, H=64, W=96), dict(quality=7), True, True),
    ("tiny_r7", dict(kind="randint", seed=11, H=64, W=96), dict(rank=7), True, True),
    ("tiny_q20", dict(kind="randint", seed=11, H=64, W=96), dict(quality=20), True, False),
    ("tiny_rank2", dict(kind="randint", seed=11, H=64, W=96), dict(rank=2), True, True),
    ("tiny_rank1", dict(kind="randint", seed=11, H=64, W=96), dict(rank=1), True, True),
    ("tiny_it0", dict(kind="randint", seed=11, H=64, W=96), dict(quality=7, num_iters=0), True, False),
    ("tiny_it1", dict(kind="randint", seed=11, H=64, W=96), dict(quality=7, num_iters=1), True, True),
    ("tiny_it2", dict(kind="randint", seed=11, H=64, W=96), dict(rank=7, num_iters=2), True, True),
    ("odd_q7", dict(kind="randint", seed=12, H=173, W=264), dict(quality=7), True, True),
    ("odd_r7", dict(kind="randint", seed=12, H=173, W=264), dict(rank=7), True, True),
    ("smooth_q7", dict(kind="smooth", seed=13, H=256, W=384), dict(quality=7), True, False),
    ("smooth_r7", dict(kind="smooth", seed=13, H=256, W=384), dict(rank=7), True, False),
    ("zero_q7", dict(kind="const", value=0, H=32, W=48), dict(quality=7), True, False),
    ("const_q7", dict(kind="const", value=100, H=32, W=48), dict(quality=7), True, False),
    ("s1_q7", dict(kind="randint", seed=0, H=512, W=768), dict(quality=7), False, False),
    ("s1_r7", dict(kind="randint", seed=0, H=512, W=768), dict(rank=7), False, False),
    ("nat_q7", dict(kind="natural"), dict(quality=7), True, False),
    ("nat_r7", dict(kind="natural"), dict(rank=7), False, False),
    ("s2odd_q7", dict(kind="randint", seed=14, H=341, W=512), dict(quality=7), False, False),
]


def main():
    torch.set_num_threads(1)
    ns = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for name, spec, kw, store_image, store_init in CASES:
        img = make_image(spec)
        enc = ns.cqmf.qmf_encode(img, **kw)
        dec = ns.cqmf.qmf_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()          # lrf/utils/metrics.py:57-71
        bpp = len(enc) * 8 / (img.shape[-2] * img.shape[-1])               # lrf/utils/metrics.py:149-162
        cr = img.numel() * img.element_size() / len(enc)                   # lrf/utils/metrics.py:120-133
        # the reference's init signs (and small init factors) per plane
        ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
        chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=tuple(kw.get("scale_factor", (0.5, 0.5))), mode="area")
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        arrays = dict(encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr), bpp=np.float64(bpp),
                      cr=np.float64(cr), spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)),
                      image_sha256=np.array(hashlib.sha256(img.numpy().tobytes()).hexdigest()),
                      decoded_sha256=np.array(hashlib.sha256(dec.numpy().tobytes()).hexdigest()),
                      ranks=np.array(meta["rank"], np.int32))
        for c, ch in enumerate(chans):
            x = ns.cqmf.patchify(ns.cutils.pad_image(ch, (8, 8), mode="reflect"), (8, 8))
            R = meta["rank"][c]
            u0, v0, _ = ns.fqmf.SVDInit(rank=R)(x.unsqueeze(0).float())
            arrays[f"sign{c}"] = wsign(v0[0].numpy())
            if store_init:
                arrays[f"u0_{c}"] = u0[0].numpy()
                arrays[f"v0_{c}"] = v0[0].numpy()
        if store_image:
            arrays["image"] = img.numpy()
            arrays["decoded"] = dec.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(spec=spec, kwargs=kw, bytes=len(enc), psnr=psnr, bpp=bpp,
                           enc_sha256=hashlib.sha256(enc).hexdigest()[:16], ranks=meta["rank"])
        print(name, index[name])

    # SVD baseline (lrf.svd_encode defaults: RGB, uint8-quantised factors)
    for name, spec, kw in [("svd_tiny_q2p5", dict(kind="randint", seed=11, H=64, W=96), dict(quality=2.5)),
                           ("svd_smooth_q2p5", dict(kind="smooth", seed=13, H=256, W=384), dict(quality=2.5)),
                           ("svd_s1_q2p5", dict(kind="randint", seed=0, H=512, W=768), dict(quality=2.5))]:
        img = make_image(spec)
        enc = ns.csvd.svd_encode(img, **kw)
        dec = ns.csvd.svd_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        x = ns.csvd.patchify(ns.cutils.pad_image(img.float(), (8, 8), mode="reflect"), (8, 8))
        s = torch.linalg.svdvals(x.double())
        arrays = dict(encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr),
                      bpp=np.float64(len(enc) * 8 / (img.shape[-2] * img.shape[-1])),
                      spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)),
                      image_sha256=np.array(hashlib.sha256(img.numpy().tobytes()).hexdigest()),
                      decoded_sha256=np.array(hashlib.sha256(dec.numpy().tobytes()).hexdigest()),
                      quant_u=np.array(meta["quantization"]["u"], np.float64),
                      quant_v=np.array(meta["quantization"]["v"], np.float64),
                      singular_values=s[:16].numpy())
        if spec["H"] <= 256:
            arrays["image"] = img.numpy()
            arrays["decoded"] = dec.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(spec=spec, kwargs=kw, bytes=len(enc), psnr=psnr, quant=meta["quantization"])
        print(name, index[name])
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def gen_sweep():
    """R-D sweep fixture (experiments/comparison/eval.py:83-100 protocol): reference (bytes, bpp, PSNR) per quality."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    spec = dict(kind="smooth", seed=21, H=128, W=192)
    img = make_image(spec)
    out = {"spec": spec, "records": []}
    for q in (1, 2, 3.5, 5, 7, 10, 12.5, 15, 20, 25, 32, 40, 60):
        enc = ns.cqmf.qmf_encode(img, quality=q)
        dec = ns.cqmf.qmf_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        out["records"].append({"quality": q, "bytes": len(enc), "bpp": len(enc) * 8 / (img.shape[-2] * img.shape[-1]),
                               "psnr": (20 * torch.log10(255 / torch.sqrt(mse))).item(), "ranks": meta["rank"]})
    out["image_sha256"] = hashlib.sha256(img.numpy().tobytes()).hexdigest()
    with open(os.path.join(OUT, "sweep_smooth.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["records"])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    gen_sweep()


def config3_image(idx, natural=None):
    """Image idx (0..23) of the BASELINE config-3 stand-in set (the Kodak files are not in the reference repository): twenty
    smooth synthetic 512x768 images (seeds 100..119) and four 512x768 crops of the reference's natural README figure.
    tests/conftest.py carries the same recipe."""
    if idx < 20:
        return make_image(dict(kind="smooth", seed=100 + idx, H=512, W=768))
    nat = make_image(dict(kind="natural")) if natural is None else natural
    y0, x0 = ((0, 0), (150, 0), (0, 224), (150, 224))[idx - 20]
    return nat[:, y0:y0 + 512, x0:x0 + 768].contiguous()


def gen_sweep512():
    """BASELINE config 3 at its stated size: the reference's (bytes, bpp, PSNR, ranks) for quality 1..32 on three of the 24
    512x768 images of the stand-in set (one thread, like every fixture)."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    out = {"images": []}
    for idx in (0, 10, 20):
        img = config3_image(idx)
        recs = []
        for q in range(1, 33):
            enc = ns.cqmf.qmf_encode(img, quality=q)
            dec = ns.cqmf.qmf_decode(enc)
            mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
            meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
            recs.append({"quality": q, "bytes": len(enc), "bpp": len(enc) * 8 / (img.shape[-2] * img.shape[-1]),
                         "psnr": (20 * torch.log10(255 / torch.sqrt(mse))).item(), "ranks": meta["rank"]})
            print(idx, recs[-1], flush=True)
        out["images"].append({"index": idx, "image_sha256": hashlib.sha256(img.numpy().tobytes()).hexdigest(), "records": recs})
    with open(os.path.join(OUT, "sweep512.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep512":
    gen_sweep512()


RGBSPACE_CASES = [
    # qmf_encode(color_space="RGB"): name, image spec, encoder kwargs, store_image
    ("rgbsp_tiny_q4", dict(kind="randint", seed=11, H=64, W=96), dict(quality=4.0), True),
    ("rgbsp_tiny_r1", dict(kind="randint", seed=11, H=64, W=96), dict(rank=1), True),
    ("rgbsp_tiny_r3_it2", dict(kind="randint", seed=11, H=64, W=96), dict(rank=3, num_iters=2), True),
    ("rgbsp_odd_q6", dict(kind="randint", seed=12, H=173, W=264), dict(quality=6.0), True),
    ("rgbsp_smooth_q2", dict(kind="smooth", seed=13, H=256, W=384), dict(quality=2.0), True),
    ("rgbsp_smooth_q10", dict(kind="smooth", seed=13, H=256, W=384), dict(quality=10.0), True),
    ("rgbsp_nat_q5", dict(kind="natural"), dict(quality=5.0), False),
]


def gen_rgbspace():
    """Fixtures of the RGB colour-space branch (lrf/compression/qmf.py:164-187): bytes, decoded image, PSNR and the
    reference's initial factors (u0, v0) so that the BCD can be checked bit for bit from the same start."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    index = {}
    for name, spec, kw, store_image in RGBSPACE_CASES:
        img = make_image(spec)
        enc = ns.cqmf.qmf_encode(img, color_space="RGB", **kw)
        dec = ns.cqmf.qmf_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        x = ns.cqmf.patchify(ns.cutils.pad_image(img.float(), (8, 8), mode="reflect"), (8, 8))
        R = meta["rank"]
        u0, v0, _ = ns.fqmf.SVDInit(rank=R)(x.unsqueeze(0).float())
        arrays = dict(encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr),
                      bpp=np.float64(len(enc) * 8 / (img.shape[-2] * img.shape[-1])),
                      spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)),
                      image_sha256=np.array(hashlib.sha256(img.numpy().tobytes()).hexdigest()),
                      decoded_sha256=np.array(hashlib.sha256(dec.numpy().tobytes()).hexdigest()),
                      rank=np.int32(R), sign=wsign(v0[0].numpy()),
                      u0=np.ascontiguousarray(u0[0].numpy()), v0=np.ascontiguousarray(v0[0].numpy()))
        if store_image:
            arrays["image"] = img.numpy()
            arrays["decoded"] = dec.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(spec=spec, kwargs=kw, bytes=len(enc), psnr=psnr, rank=R,
                           enc_sha256=hashlib.sha256(enc).hexdigest()[:16])
        print(name, index[name])
    with open(os.path.join(OUT, "index_rgbspace.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


RGBSPACE_ANY_CASES = [
    # qmf_encode(color_space="RGB") beyond 8x8 patches: name, image spec, encoder kwargs
    ("rgbany_p4_q6", dict(kind="randint", seed=31, H=50, W=70), dict(quality=6.0, patch_size=(4, 4))),
    ("rgbany_p16_r5", dict(kind="smooth", seed=32, H=96, W=144), dict(rank=5, patch_size=(16, 16))),
    ("rgbany_p8x4_q3", dict(kind="randint", seed=33, H=61, W=45), dict(quality=3.0, patch_size=(8, 4), num_iters=3)),
    ("rgbany_nopatch_q8", dict(kind="smooth", seed=34, H=64, W=96), dict(quality=8.0, patch=False)),
    ("rgbany_nopatch_r2", dict(kind="randint", seed=35, H=37, W=53), dict(rank=2, patch=False, num_iters=2)),
    ("rgbany_p8_it0", dict(kind="smooth", seed=36, H=64, W=96), dict(quality=4.0, num_iters=0)),
]


def gen_rgbspace_any():
    """The RGB colour-space branch for other patch sizes, patch=False and num_iters=0 (lrf/compression/qmf.py:164-212):
    bytes, decoded image, PSNR and the reference's initial factors (from which the BCD is bit-reproducible)."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    index = {}
    for name, spec, kw in RGBSPACE_ANY_CASES:
        img = make_image(spec)
        enc = ns.cqmf.qmf_encode(img, color_space="RGB", **kw)
        dec = ns.cqmf.qmf_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        R = meta["rank"]
        if kw.get("patch", True):
            ps = tuple(kw.get("patch_size", (8, 8)))
            x = ns.cqmf.patchify(ns.cutils.pad_image(img.float(), ps, mode="reflect"), ps).unsqueeze(0)
        else:
            x = img.float()
        u0, v0, _ = ns.fqmf.SVDInit(rank=R)(x)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr),
                            spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)), image=img.numpy(), decoded=dec.numpy(),
                            rank=np.int32(R), u0=np.ascontiguousarray(u0.numpy()), v0=np.ascontiguousarray(v0.numpy()))
        index[name] = dict(bytes=len(enc), psnr=psnr, rank=R)
        print(name, index[name], flush=True)
    with open(os.path.join(OUT, "index_rgbspace_any.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rgbspace_any":
    gen_rgbspace_any()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rgbspace":
    gen_rgbspace()


ANYSHAPE_CASES = [
    # qmf_encode(color_space="YCbCr") with other patch sizes / patch=False (experiments/ablation_patchsize/eval.py:49-55):
    # name, image spec, encoder kwargs, store_image, store_init
    ("any_p4_q20", dict(kind="randint", seed=21, H=64, W=96), dict(quality=20, patch_size=(4, 4)), True, True),
    ("any_p4_odd_q40", dict(kind="smooth", seed=22, H=72, W=104), dict(quality=40, patch_size=(4, 4)), True, True),
    ("any_p16_q10", dict(kind="smooth", seed=23, H=72, W=104), dict(quality=10, patch_size=(16, 16)), True, True),
    ("any_p16_r70", dict(kind="randint", seed=24, H=128, W=192), dict(rank=70, patch_size=(16, 16)), True, False),
    ("any_p32_q30", dict(kind="randint", seed=12, H=173, W=264), dict(quality=30, patch_size=(32, 32)), True, True),
    ("any_p8x4_q15", dict(kind="smooth", seed=25, H=64, W=96), dict(quality=15, patch_size=(8, 4)), True, True),
    ("any_nopatch_q10", dict(kind="randint", seed=26, H=64, W=96), dict(quality=10, patch=False), True, True),
    ("any_nopatch_odd_q25", dict(kind="smooth", seed=27, H=75, W=101), dict(quality=25, patch=False), True, True),
    ("any_nopatch_it1", dict(kind="smooth", seed=27, H=75, W=101), dict(quality=25, patch=False, num_iters=1), True, True),
    ("any_nopatch_wide_bounds", dict(kind="smooth", seed=28, H=64, W=96), dict(quality=8, patch=False, bounds=(-128, 127)), True, True),
    ("any_s1_p4_q40", dict(kind="smooth", seed=5, H=512, W=768), dict(quality=40, patch_size=(4, 4)), False, False),
    ("any_s1_p16_q20", dict(kind="smooth", seed=5, H=512, W=768), dict(quality=20, patch_size=(16, 16)), False, False),
    ("any_s1_p32_q20", dict(kind="smooth", seed=5, H=512, W=768), dict(quality=20, patch_size=(32, 32)), False, False),
    ("any_s1_nopatch_q20", dict(kind="smooth", seed=5, H=512, W=768), dict(quality=20, patch=False), False, False),
]
SCALE_CASES = [
    # chroma scale factors other than (0.5, 0.5) (lrf/compression/qmf.py:230): 8x8 patches, another patch size, no patches
    ("sf_quarter_q10", dict(kind="smooth", seed=51, H=64, W=96), dict(quality=10, scale_factor=(0.25, 0.25)), True, True),
    ("sf_444_r5", dict(kind="randint", seed=52, H=40, W=56), dict(rank=5, scale_factor=(1.0, 1.0)), True, True),
    ("sf_mixed_odd_q12", dict(kind="smooth", seed=53, H=75, W=101), dict(quality=12, scale_factor=(0.3, 0.7), patch_size=(4, 4)), True, True),
    ("sf_nopatch_q8", dict(kind="smooth", seed=54, H=64, W=96), dict(quality=8, scale_factor=(0.5, 0.25), patch=False), True, True),
]


def gen_anyshape(cases=None, index_name="index_anyshape.json"):
    """Fixtures of the patch-size / patch=False branches: bytes, decoded image, PSNR, the reference's per-plane matrices'
    initial factors (u0, v0) for the small cases and their column signs for all."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    index = {}
    for name, spec, kw, store_image, store_init in (ANYSHAPE_CASES if cases is None else cases):
        img = make_image(spec)
        enc = ns.cqmf.qmf_encode(img, **kw)
        dec = ns.cqmf.qmf_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
        chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=tuple(kw.get("scale_factor", (0.5, 0.5))), mode="area")
        arrays = dict(encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr),
                      bpp=np.float64(len(enc) * 8 / (img.shape[-2] * img.shape[-1])),
                      spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)),
                      image_sha256=np.array(hashlib.sha256(img.numpy().tobytes()).hexdigest()),
                      decoded_sha256=np.array(hashlib.sha256(dec.numpy().tobytes()).hexdigest()),
                      ranks=np.array(meta["rank"], np.int32))
        for c, ch in enumerate(chans):
            if kw.get("patch", True):
                ps = kw.get("patch_size", (8, 8))
                x = ns.cqmf.patchify(ns.cutils.pad_image(ch, ps, mode="reflect"), ps)
            else:
                x = ch[0]
            R = meta["rank"][c]
            u0, v0, _ = ns.fqmf.SVDInit(rank=R)(x.unsqueeze(0).float())
            arrays[f"sign{c}"] = wsign(v0[0].numpy())
            if store_init:
                arrays[f"u0_{c}"] = u0[0].numpy()
                arrays[f"v0_{c}"] = v0[0].numpy()
        if store_image:
            arrays["image"] = img.numpy()
            arrays["decoded"] = dec.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(spec=spec, kwargs=kw, bytes=len(enc), psnr=psnr, ranks=meta["rank"],
                           enc_sha256=hashlib.sha256(enc).hexdigest()[:16])
        print(name, index[name], flush=True)
    with open(os.path.join(OUT, index_name), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "scale":
    gen_anyshape(SCALE_CASES, "index_scale.json")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "anyshape":
    gen_anyshape()


QMFX_R3 = [  # round 3: SVDInit(num_levels=...) (qmf.py:56-68) and CoordinateDescent(eps=...)
    ("qmfx_levels_f01", dict(seed=9, M=300, N=64), dict(rank=4, num_iters=5, bounds=(-16, 15), factor=(0, 1), num_levels=31)),
    ("qmfx_levels_f012", dict(seed=10, M=128, N=96), dict(rank=3, num_iters=4, num_levels=15.0)),
    ("qmfx_levels_it0", dict(seed=12, M=96, N=64), dict(rank=3, num_iters=0, num_levels=7)),
    ("qmfx_eps", dict(seed=11, M=200, N=64), dict(rank=5, num_iters=3, bounds=(-8, 7), factor=(0, 1), eps=1e-3)),
]


def gen_qmfx(cases=None, index_name="index_qmfx.json"):
    """The QMF class beyond what qmf_encode uses (lrf/factorization/qmf.py:74-231): unbounded factors, elastic-net terms,
    the affine pair w (factor containing 2) — on the shape of the reference's own smoke test (test/test_factorization.py:5-10:
    randint(0, 256, (1, 784, 192)), rank 5, 10 iterations) and a small bounded case.  Stored: the reference's initial factors
    (its LAPACK SVD), its final (u, v, w) and loss."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    QMF = ns.fqmf.QMF
    cases = cases or [
        ("qmfx_unbounded_f01", dict(seed=5, M=784, N=192), dict(rank=5, num_iters=10, factor=(0, 1))),
        ("qmfx_unbounded_f012", dict(seed=5, M=784, N=192), dict(rank=5, num_iters=10)),  # the reference's smoke test, as is
        ("qmfx_bounded_l2", dict(seed=6, M=300, N=64), dict(rank=4, num_iters=5, bounds=(-16, 15), factor=(0, 1), l2=(0.02, 0.01), l1_ratio=0.5)),
        ("qmfx_unbounded_l2_f012", dict(seed=7, M=128, N=96), dict(rank=3, num_iters=4, l2=0.001, l1_ratio=0.25)),
        ("qmfx_bounded_f0", dict(seed=8, M=200, N=64), dict(rank=6, num_iters=3, bounds=(-8, 7), factor=(0,))),
    ]
    index = {}
    for name, spec, kw in cases:
        g = torch.Generator().manual_seed(spec["seed"])
        x = torch.randint(0, 256, (1, spec["M"], spec["N"]), generator=g).float()
        qmf = QMF(**kw)
        u0, v0, w0 = qmf.init(x)
        u, v, w = qmf.decompose(x)
        loss = QMF.loss(x, u, v, w)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), spec=json.dumps(spec), kwargs=json.dumps(kw), u0=u0[0].numpy(), v0=v0[0].numpy(),
                            u=u[0].numpy(), v=v[0].numpy(), w=w[0].numpy().reshape(2), loss=np.float64(loss.item()),
                            sign=wsign(v0[0].numpy()), w0=w0[0].numpy().reshape(2))
        index[name] = {"loss": loss.item(), "w": w[0].reshape(2).tolist()}
        print(name, index[name], flush=True)
    with open(os.path.join(OUT, index_name), "w") as f:
        json.dump(index, f, indent=1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "qmfx":
    gen_qmfx()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "qmfx_r3":
    gen_qmfx(QMFX_R3, "index_qmfx_r3.json")


SVD_ANY_CASES = [
    # svd_encode's RGB branch beyond the default: name, image spec, kwargs (dtype by name)
    ("svdany_p4_q5", dict(kind="smooth", seed=41, H=50, W=70), dict(quality=5.0, patch_size=(4, 4))),
    ("svdany_p16_r6", dict(kind="smooth", seed=42, H=96, W=144), dict(rank=6, patch_size=(16, 16))),
    ("svdany_nopatch_q6", dict(kind="smooth", seed=43, H=64, W=96), dict(quality=6.0, patch=False)),
    ("svdany_p8_float", dict(kind="smooth", seed=44, H=64, W=96), dict(quality=3.0, dtype="float32")),
    ("svdany_nopatch_float", dict(kind="randint", seed=45, H=37, W=53), dict(rank=3, patch=False, dtype="float32")),
]


def gen_svd_any():
    """svd_encode / svd_decode RGB branch with other patch sizes, patch=False and float factors (lrf/compression/svd.py:157-193,
    310-326): reference bytes, decoded pixels, PSNR."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    import importlib
    csvd = importlib.import_module("lrf.compression.svd")
    index = {}
    for name, spec, kw in SVD_ANY_CASES:
        img = make_image(spec)
        kwr = dict(kw)
        if "dtype" in kwr:
            kwr["dtype"] = getattr(torch, kwr["dtype"])
        enc = csvd.svd_encode(img, **kwr)
        dec = csvd.svd_decode(enc)
        mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
        psnr = (20 * torch.log10(255 / torch.sqrt(mse))).item()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr),
                            spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)), image=img.numpy(), decoded=dec.numpy())
        index[name] = dict(bytes=len(enc), psnr=psnr)
        print(name, index[name], flush=True)
    with open(os.path.join(OUT, "index_svd_any.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "svd_any":
    gen_svd_any()


def gen_loess():
    """LOESS fixture (lrf/utils/misc.py:276-412): the reference's class is taken out of its module by name (the module
    itself needs seaborn / pyinstrument, absent here) and run on seeded samples; inputs and predictions are stored."""
    import ast
    from itertools import product  # noqa: F401  (names the class body uses)
    from typing import Optional, Sequence  # noqa: F401
    from scipy.linalg import lstsq  # noqa: F401
    path = os.path.join(ref_loader.REF_ROOT, "lrf", "utils", "misc.py")
    tree = ast.parse(open(path).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LOESS"][0]
    scope = dict(np=np, product=product, Optional=Optional, Sequence=Sequence, lstsq=lstsq)
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scope)
    LOESS = scope["LOESS"]
    rng = np.random.default_rng(7)
    out = {}
    for name, n in (("a", 40), ("b", 25)):
        x = np.sort(rng.uniform(0.05, 1.2, n))
        y = 20 + 12 * np.log1p(4 * x) + rng.normal(0, 0.4, n)
        grid = np.linspace(0.0, 1.3, 27)
        single = LOESS(frac=0.3, degree=1).fit(x, y).predict(grid)
        model = LOESS(frac=np.arange(0.15, 0.75, 0.1), degree=[1, 2]).fit(x, y)
        out[name] = dict(x=x.tolist(), y=y.tolist(), grid=grid.tolist(), single=single.tolist(),
                         searched=model.predict(grid).tolist(), best_frac=float(model.best_frac), best_degree=int(model.best_degree))
    with open(os.path.join(OUT, "loess.json"), "w") as f:
        json.dump(out, f)
    print({k: (v["best_frac"], v["best_degree"]) for k, v in out.items()})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "loess":
    gen_loess()


# ---- the reference at 8 torch threads against itself at one thread ---------------------------------
# MKL splits the long `x.mT @ u` reduction (lrf/factorization/qmf.py:107 via :139) differently with more threads, so the
# reference's own byte stream depends on torch.get_num_threads().  Every other fixture is a one-thread run; this one records
# how far a default multi-threaded reference user lands from it: stream size, bpp, PSNR and the fraction of int8 factor
# entries that differ, per case; where the two streams differ (and the image is Kodak-sized or smaller) both are kept, so
# that a test can measure the distance of the oracle / HIP result to either directly.
THREAD_CASES = [
    ("thr_s1_q7", dict(kind="randint", seed=0, H=512, W=768), dict(quality=7)),
    ("thr_s1_r7", dict(kind="randint", seed=0, H=512, W=768), dict(rank=7)),
    ("thr_s3_r7", dict(kind="randint", seed=3, H=512, W=768), dict(rank=7)),
    ("thr_smooth_r7", dict(kind="smooth", seed=21, H=512, W=768), dict(rank=7)),
    ("thr_smooth_q20", dict(kind="smooth", seed=22, H=512, W=768), dict(quality=20)),
    ("thr_nat_r7", dict(kind="natural"), dict(rank=7)),
    ("thr_odd_r7", dict(kind="randint", seed=12, H=173, W=264), dict(rank=7)),
    ("thr_clic_q7", dict(kind="randint", seed=5, H=1365, W=2048), dict(quality=7)),
    ("thr_clic_r7", dict(kind="randint", seed=6, H=1365, W=2048), dict(rank=7)),
    ("thr_clic_smooth_r7", dict(kind="smooth", seed=23, H=1365, W=2048), dict(rank=7)),
]


def gen_threads(n_threads=8):
    ns = ref_loader.load()
    records, streams = [], {}
    for name, spec, kw in THREAD_CASES:
        torch.set_num_threads(1)  # (the "smooth" recipe itself is pinned at one thread)
        img = make_image(spec)
        out = {}
        for nt in (1, n_threads):
            torch.set_num_threads(nt)
            enc = ns.cqmf.qmf_encode(img, **kw)
            dec = ns.cqmf.qmf_decode(enc)
            mse = torch.mean((img.float() - dec.float()) ** 2, dim=(-3, -2, -1))
            fac = [ns.cutils.decode_tensor(f).numpy() for f in ns.cutils.separate_bytes(ns.cutils.separate_bytes(enc, 2)[1], 6)]
            out[nt] = dict(enc=enc, fac=fac, len=len(enc), bpp=len(enc) * 8 / (img.shape[-2] * img.shape[-1]),
                           psnr=(20 * torch.log10(255 / torch.sqrt(mse))).item(), sha256=hashlib.sha256(enc).hexdigest())
        torch.set_num_threads(1)
        ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
        chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=(0.5, 0.5), mode="area")
        meta = json.loads(ns.cutils.separate_bytes(out[1]["enc"], 2)[0].decode())
        signs = []
        for c, ch in enumerate(chans):
            x = ns.cqmf.patchify(ns.cutils.pad_image(ch, (8, 8), mode="reflect"), (8, 8))
            u0, v0, _ = ns.fqmf.SVDInit(rank=meta["rank"][c])(x.unsqueeze(0).float())
            signs.append([int(s) for s in wsign(v0[0].numpy())])
        a, b = out[1], out[n_threads]
        ndiff = sum(int((x != y).sum()) for x, y in zip(a["fac"], b["fac"]))
        ntot = sum(x.size for x in a["fac"])
        rec = dict(name=name, spec=spec, kwargs=kw, ranks=meta["rank"], signs=signs, threads=n_threads,
                   image_sha256=hashlib.sha256(img.numpy().tobytes()).hexdigest(),
                   t1={k: a[k] for k in ("len", "bpp", "psnr", "sha256")}, tN={k: b[k] for k in ("len", "bpp", "psnr", "sha256")},
                   differing_entries=ndiff, entries=ntot,
                   max_abs_entry_diff=max(int(np.abs(x.astype(np.int16) - y.astype(np.int16)).max()) for x, y in zip(a["fac"], b["fac"])))
        records.append(rec)
        if img.shape[-2] * img.shape[-1] <= 512 * 768 * 2 and a["sha256"] != b["sha256"]:
            streams[name + "_t1"] = np.frombuffer(a["enc"], np.uint8)
            streams[name + "_tN"] = np.frombuffer(b["enc"], np.uint8)
        print(name, {k: rec[k] for k in ("t1", "tN", "differing_entries", "entries", "max_abs_entry_diff")}, flush=True)
    with open(os.path.join(OUT, "threads8.json"), "w") as f:
        json.dump(records, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "threads8_streams.npz"), **streams)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "threads":
    gen_threads()


# ---- the byte-identity RATE ------------------------------------------------------------------------
# "With the reference's LAPACK column signs the encoder emits the reference's byte stream" was shown on 14 fixtures; this
# measures how often that holds on a population: >= 100 (image, parameters) cases — the 24 config-3 stand-in images
# (smooth + crops of the natural figure) at rank 7 / quality 20 / quality 32, random and other smooth 512x768 images, a few of
# 1365x2048 — the 1-thread reference against the oracle with the reference's signs: identical streams, or how far apart
# (factor entries, PSNR, stream size).  Writes tests/golden/identity_rate.json (records + summary; no streams kept).
def identity_cases():
    cases = []
    for i in range(24):
        spec = dict(kind="config3", idx=i)
        cases += [(f"c3_{i:02d}_r7", spec, dict(rank=7)), (f"c3_{i:02d}_q20", spec, dict(quality=20)), (f"c3_{i:02d}_q32", spec, dict(quality=32))]
    for k in range(20):
        cases.append((f"rnd_{k:02d}", dict(kind="randint", seed=100 + k, H=512, W=768), dict(rank=7) if k % 2 == 0 else dict(quality=7)))
    for k in range(6):
        cases.append((f"smooth_{k}", dict(kind="smooth", seed=300 + k, H=512, W=768), dict(quality=7)))
    cases += [("clic_rnd_0", dict(kind="randint", seed=400, H=1365, W=2048), dict(rank=7)),
              ("clic_rnd_1", dict(kind="randint", seed=401, H=1365, W=2048), dict(quality=7)),
              ("clic_smooth_0", dict(kind="smooth", seed=402, H=1365, W=2048), dict(rank=7)),
              ("clic_smooth_1", dict(kind="smooth", seed=403, H=1365, W=2048), dict(quality=20))]
    return cases


def identity_image(spec, natural=None):
    if spec["kind"] == "config3":
        return config3_image(spec["idx"], natural)
    return make_image(spec)


def gen_identity():
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle import oracle
    from lrf_amd.codec import pack_image  # host-side container code only: no GPU call
    ns = ref_loader.load()
    torch.set_num_threads(1)
    natural = make_image(dict(kind="natural"))
    records = []
    for name, spec, kw in identity_cases():
        img = identity_image(spec, natural)
        enc = ns.cqmf.qmf_encode(img, **kw)
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        ref_fac = [ns.cutils.decode_tensor(f).numpy() for f in ns.cutils.separate_bytes(ns.cutils.separate_bytes(enc, 2)[1], 6)]
        dec = ns.cqmf.qmf_decode(enc)
        psnr_ref = (20 * torch.log10(255 / torch.sqrt(torch.mean((img.float() - dec.float()) ** 2)))).item()
        ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
        chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=(0.5, 0.5), mode="area")
        X = oracle.rgb_to_planes(img.numpy())
        fac, signs = [], []
        for c, ch in enumerate(chans):
            x = ns.cqmf.patchify(ns.cutils.pad_image(ch, (8, 8), mode="reflect"), (8, 8))
            _, v0, _ = ns.fqmf.SVDInit(rank=meta["rank"][c])(x.unsqueeze(0).float())
            signs.append([int(v) for v in wsign(v0[0].numpy())])
            u, v = oracle.qmf_decompose(X[c], meta["rank"][c], 10, (-16, 15), sign=np.array(signs[-1], np.int8))
            fac += [u.astype(np.int8), v.astype(np.int8)]
        H, W = img.shape[-2:]
        stream = pack_image(fac, (H, W), meta["rank"], (-16, 15), (8, 8), "uint8")
        ident = stream == enc
        ndiff = sum(int((a != b).sum()) for a, b in zip(fac, ref_fac))
        ntot = sum(a.size for a in ref_fac)
        out = oracle.planes_to_rgb(fac[0::2], fac[1::2], H, W)
        psnr_or = float(20 * np.log10(255 / np.sqrt(np.mean((img.numpy().astype(np.float32) - out.astype(np.float32)) ** 2))))
        rec = dict(name=name, spec=spec, kwargs=kw, ranks=meta["rank"], signs=signs, identical=bool(ident), differing_entries=ndiff, entries=ntot,
                   ref_len=len(enc), oracle_len=len(stream), ref_psnr=psnr_ref, oracle_psnr=psnr_or,
                   ref_sha256=hashlib.sha256(enc).hexdigest(), oracle_sha256=hashlib.sha256(stream).hexdigest())
        records.append(rec)
        print(name, "identical" if ident else f"DIFF entries {ndiff}/{ntot} = {ndiff / ntot:.4f}, dPSNR {psnr_or - psnr_ref:+.4f} dB, "
              f"size {len(stream)} vs {len(enc)}", flush=True)
    n = len(records)
    nid = sum(r["identical"] for r in records)
    diff = [r for r in records if not r["identical"]]
    summary = dict(cases=n, identical=nid, rate=nid / n,
                   max_entry_fraction=max((r["differing_entries"] / r["entries"] for r in diff), default=0.0),
                   max_abs_psnr_diff_db=max((abs(r["oracle_psnr"] - r["ref_psnr"]) for r in diff), default=0.0),
                   max_size_fraction=max((abs(r["oracle_len"] - r["ref_len"]) / r["ref_len"] for r in diff), default=0.0),
                   mean_psnr_diff_db=float(np.mean([r["oracle_psnr"] - r["ref_psnr"] for r in records])),
                   mean_size_ratio=float(np.mean([r["oracle_len"] / r["ref_len"] for r in records])),
                   by_params={})
    for key in ("rank=7", "quality=7", "quality=20", "quality=32"):
        k, v = key.split("=")
        sel = [r for r in records if r["kwargs"].get(k) == int(v)]
        if sel:
            summary["by_params"][key] = dict(cases=len(sel), identical=sum(r["identical"] for r in sel))
    with open(os.path.join(OUT, "identity_rate.json"), "w") as f:
        json.dump(dict(what="1-thread reference qmf_encode vs the oracle with the reference's LAPACK column signs (tools/gen_golden.py identity)",
                       torch=torch.__version__, summary=summary, records=records), f, indent=1)
    print(summary)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "identity":
    gen_identity()


# ---- qmf_encode(**kwargs) reaching the general solver (lrf/compression/qmf.py:127, 256) ---------------
QMFKW_CASES = [
    # name, image spec, encoder kwargs (l2 / l1_ratio / eps / num_levels are forwarded to QMF(...))
    ("kw_l2_q20", dict(kind="smooth", seed=51, H=64, W=96), dict(quality=20, l2=0.5, l1_ratio=0.3)),
    ("kw_levels_r3", dict(kind="smooth", seed=52, H=64, W=96), dict(rank=3, num_levels=12)),
    ("kw_nopatch_l2", dict(kind="smooth", seed=54, H=40, W=56), dict(quality=15, patch=False, l2=(1.0, 0.25))),
    ("kw_p4_eps", dict(kind="randint", seed=55, H=48, W=64), dict(quality=30, patch_size=(4, 4), eps=1e-3, l2=0.05)),
    ("kw_rgb_l2", dict(kind="randint", seed=53, H=48, W=64), dict(rank=4, color_space="RGB", l2=(0.1, 0.2), l1_ratio=0.5)),
]


def gen_qmfkw():
    """Stored per case: the reference's stream, decoded image hash, PSNR, and per matrix its initial factors (u0, v0, w0) as
    SVDInit(rank, num_levels) returns them (scaled when num_levels is given)."""
    torch.set_num_threads(1)
    ns = ref_loader.load()
    index = {}
    for name, spec, kw in QMFKW_CASES:
        img = make_image(spec)
        enc = ns.cqmf.qmf_encode(img, **kw)
        dec = ns.cqmf.qmf_decode(enc)
        psnr = (20 * torch.log10(255 / torch.sqrt(torch.mean((img.float() - dec.float()) ** 2)))).item()
        meta = json.loads(ns.cutils.separate_bytes(enc, 2)[0].decode())
        arrays = dict(encoded=np.frombuffer(enc, np.uint8), psnr=np.float64(psnr), spec=np.array(json.dumps(spec)), kwargs=np.array(json.dumps(kw)),
                      image=img.numpy(), decoded_sha256=np.array(hashlib.sha256(dec.numpy().tobytes()).hexdigest()))
        ps = tuple(kw.get("patch_size", (8, 8)))
        if kw.get("color_space", "YCbCr") == "RGB":
            mats = [ns.cqmf.patchify(ns.cutils.pad_image(img.float(), ps, mode="reflect"), ps)]
            ranks = [meta["rank"]]
        else:
            ycbcr = ns.cutils.rgb_to_ycbcr(img.float())
            chans = ns.cutils.chroma_downsampling(ycbcr, scale_factor=(0.5, 0.5), mode="area")
            mats = [ns.cqmf.patchify(ns.cutils.pad_image(ch, ps, mode="reflect"), ps) if kw.get("patch", True) else ch[0] for ch in chans]
            ranks = meta["rank"]
        for c, (x, R) in enumerate(zip(mats, ranks)):
            u0, v0, w0 = ns.fqmf.SVDInit(rank=R, num_levels=kw.get("num_levels"))(x.unsqueeze(0).float())
            arrays[f"u0_{c}"], arrays[f"v0_{c}"], arrays[f"w0_{c}"] = u0[0].numpy(), v0[0].numpy(), w0[0].numpy().reshape(2)
            arrays[f"sign{c}"] = wsign(v0[0].numpy())
        arrays["ranks"] = np.array(ranks, np.int32)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        index[name] = dict(spec=spec, kwargs=kw, bytes=len(enc), psnr=psnr, ranks=[int(r) for r in ranks], enc_sha256=hashlib.sha256(enc).hexdigest()[:16])
        print(name, index[name], flush=True)
    with open(os.path.join(OUT, "index_qmfkw.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "qmfkw":
    gen_qmfkw()
