"""GPU suite (-m gpu): the batched R-D sweep (lrf_amd.rd_sweep_batched: all images of one size in one call per quality) against the
per-image loop the reference's experiments run (lrf_amd.rd_sweep; experiments/comparison/eval.py:83-110)."""
import pytest
import torch

from conftest import config3_image

pytestmark = pytest.mark.gpu


def test_batched_sweep_equals_the_per_image_loop():
    import lrf_amd
    imgs = [config3_image(i)[:, :96, :160].contiguous() for i in (0, 5, 20, 23)]  # smooth synthetic ones and natural crops
    qualities = (2, 9, 21, 33)  # rank families 8, 16 and 32
    one = lrf_amd.rd_sweep(imgs, qualities, lrf_amd.qmf_encode, lrf_amd.qmf_decode)
    many = lrf_amd.rd_sweep_batched(torch.stack(imgs), qualities)
    assert len(one) == len(many) == len(imgs) * len(qualities)
    key = lambda r: (r["image"], r["quality"])
    many = {key(r): r for r in many}
    for r in one:
        m = many[key(r)]
        for k in ("compression ratio", "bit rate (bpp)", "PSNR (dB)", "SSIM"):
            assert r[k] == m[k], (key(r), k, r[k], m[k])  # the same byte streams, the same pixels


def test_batched_sweep_on_another_patch_size():
    """the any-shape branches through the same sweep (qmf_encode_batch packs natively, the decoder goes stream by stream)"""
    import lrf_amd
    imgs = [config3_image(i)[:, :64, :96].contiguous() for i in (1, 22)]
    one = lrf_amd.rd_sweep(imgs, (10, 25), lrf_amd.qmf_encode, lrf_amd.qmf_decode, patch_size=(16, 16))
    many = lrf_amd.rd_sweep_batched(torch.stack(imgs), (10, 25), patch_size=(16, 16))
    key = lambda r: (r["image"], r["quality"])
    many = {key(r): r for r in many}
    for r in one:
        for k in ("bit rate (bpp)", "PSNR (dB)"):
            assert r[k] == many[key(r)][k], (key(r), k)


def test_fused_sweep_emits_the_streams_of_the_per_quality_calls():
    """lrf_qmf_encode_sweep_rgb_u8 (one GPU call for every quality of an R-D sweep: planes / Gram matrices / SVD initialisation
    once per image, the lower ranks taking the leading columns of the largest rank's initialisation, the iterations of all
    (quality, image) pairs in per-family launches) against one qmf_encode_batch call per quality: byte-identical streams —
    12 of BASELINE config 3's 512x768 images at qualities 1..32 (twenty rank triples, all three kernel families, the
    persistent kernel: 12 x 20 x 24 = 5760 blocks), and a small call that does not split its families, with sign vectors."""
    import hashlib
    import lrf_amd
    from lrf_amd.codec import qmf_ranks
    imgs = torch.stack([config3_image(i) for i in range(12)])
    qualities = list(range(1, 33))
    fused = lrf_amd.qmf_encode_sweep(imgs, qualities=qualities)
    assert len(fused) == len(qualities) and all(len(s) == imgs.shape[0] for s in fused)
    dev = imgs.cuda()
    seen = {}
    for q, streams in zip(qualities, fused):
        t = tuple(qmf_ranks(imgs.shape[-2:], None, q))
        if t not in seen:
            seen[t] = lrf_amd.qmf_encode_batch(dev, quality=q)
        assert streams == seen[t], (q, t)
    assert len(seen) >= 18
    # explicit rank triples, a call too small to split its families (3 images of 96x160), component signs
    small = torch.stack([config3_image(i)[:, :96, :160].contiguous() for i in (0, 5, 23)])
    triples = [(3, 1, 2), (12, 6, 6), (21, 10, 10), (7, 3, 3)]
    g = torch.Generator().manual_seed(3)
    sign = (torch.randint(0, 2, (21 + 10 + 10,), generator=g) * 2 - 1).to(torch.int8)
    fused = lrf_amd.qmf_encode_sweep(small, ranks=triples, init_sign=sign)
    for t, streams in zip(triples, fused):
        offs = (0, 21, 31)
        sg = torch.cat([sign[offs[c]:offs[c] + t[c]] for c in range(3)])
        assert streams == lrf_amd.qmf_encode_batch(small.cuda(), rank=list(t), init_sign=sg), t


def test_fused_sweep_entry_point_edges():
    """lrf_qmf_encode_sweep_rgb_u8 through the context binding: one triple = the plain call; repeated triples and a triple whose
    chroma rank is the largest of the call; an odd-sized image (padded planes, k_planes_strip); K = 1 and K = 2; what it refuses."""
    import lrf_amd
    from lrf_amd import _lib
    ctx = _lib.context(0)
    g = torch.Generator().manual_seed(11)
    for (B, H, W), triples, K in (((5, 64, 96), [(7, 3, 3)], 10), ((3, 77, 131), [(2, 5, 1), (2, 5, 1), (9, 2, 4), (1, 1, 17)], 2),
                                  ((2, 48, 48), [(6, 6, 6), (3, 3, 3)], 1)):
        base = torch.rand(B, 3, H // 4 + 1, W // 4 + 1, generator=g) * 255
        imgs = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
                + torch.randn(B, 3, H, W, generator=g) * 5).clamp(0, 255).to(torch.uint8).cuda()
        got = ctx.encode_sweep_rgb(imgs, triples, K, -16, 15)
        assert len(got) == len(triples)
        for t, (U, V) in zip(triples, got):
            Ur, Vr = lrf_amd.qmf_factorize_batch(imgs, t, num_iters=K)
            assert torch.equal(U, Ur) and torch.equal(V, Vr), ((B, H, W), t, K)
    with pytest.raises(NotImplementedError, match="one call per triple"):
        ctx.encode_sweep_rgb(imgs, [(33, 3, 3)], 10, -16, 15)  # ranks above 32 iterate on the any-shape kernels
    with pytest.raises((ValueError, NotImplementedError)):
        ctx.encode_sweep_rgb(imgs, [(3, 3, 3)], 0, -16, 15)  # K = 0: lrf_qmf_svd_init_f32
    ctx.check()
