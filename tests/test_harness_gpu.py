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
