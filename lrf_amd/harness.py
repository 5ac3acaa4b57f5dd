"""eval_compression — the reference's per-image evaluation protocol (lrf/utils/misc.py:59-121) for this path:
one encode call and one decode call, wall-clock milliseconds each, then compression ratio, bpp, PSNR and SSIM
(metrics.ssim restates scikit-image's algorithm; scikit-image itself is absent from this image)."""
import time
from typing import Callable

import numpy as np
import torch

from .metrics import bits_per_pixel, compression_ratio, psnr, ssim


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def eval_compression(image, encoder: Callable, decoder: Callable, reconstruct: bool = False, **kwargs) -> dict:
    if isinstance(image, np.ndarray):
        image = torch.tensor(image.transpose((2, 0, 1)))  # HWC -> CHW, as the reference does
    elif not isinstance(image, torch.Tensor):
        raise ValueError("Image must be a file path, numpy array, or torch tensor.")
    _sync()
    t0 = time.perf_counter()
    encoded = encoder(image, **kwargs)
    _sync()
    encoding_time = 1000 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    reconstructed = decoder(encoded)
    _sync()
    decoding_time = 1000 * (time.perf_counter() - t0)
    output = {
        "compression ratio": compression_ratio(image, encoded),
        "bit rate (bpp)": bits_per_pixel(image.shape[-2:], encoded),
        "PSNR (dB)": psnr(image, reconstructed).item(),
        "SSIM": ssim(image, reconstructed).item(),
        "encoding time (ms)": encoding_time,
        "decoding time (ms)": decoding_time,
    }
    if reconstruct:
        output["reconstructed"] = reconstructed
    return output


def rd_sweep(images, qualities, encoder: Callable, decoder: Callable, **kwargs) -> list:
    """The loop of experiments/comparison/eval.py:83-110 for one method: every image at every quality."""
    records = []
    for idx, image in enumerate(images):
        for q in qualities:
            rec = eval_compression(image, encoder, decoder, quality=float(q), **kwargs)
            rec.update({"image": idx, "quality": float(q)})
            records.append(rec)
    return records
