"""Image-compression metrics with the reference's definitions (lrf/utils/metrics.py)."""
import functools
from operator import mul

import numpy as np
import torch


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.mean((a - b) ** 2, dim=(-3, -2, -1))  # lrf/utils/metrics.py:24-35


def psnr(img1: torch.Tensor, img2: torch.Tensor, max_value: int = 255) -> torch.Tensor:
    # lrf/utils/metrics.py:57-71
    return 20 * torch.log10(max_value / torch.sqrt(mse(img1.float(), img2.float())))


def get_memory_usage(obj) -> int:
    # lrf/utils/metrics.py:94-117
    if isinstance(obj, (list, tuple, set)):
        return sum(get_memory_usage(o) for o in obj)
    if isinstance(obj, dict):
        return sum(get_memory_usage(o) for o in obj.values())
    if isinstance(obj, bytes):
        return len(obj)
    if isinstance(obj, np.ndarray):
        return obj.nbytes
    if isinstance(obj, torch.Tensor):
        return obj.numel() * obj.element_size()
    raise ValueError("Unsupported data type. Please provide an object containing NumPy arrays or PyTorch tensors.")


def compression_ratio(input, compressed) -> float:
    return get_memory_usage(input) / get_memory_usage(compressed)  # lrf/utils/metrics.py:120-133


def bits_per_pixel(size, compressed) -> float:
    # lrf/utils/metrics.py:149-162
    return get_memory_usage(compressed) * 8 / functools.reduce(mul, size, 1)


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Mean structural similarity of two (C, H, W) images, as lrf/utils/metrics.py:74-91 computes it.

    The reference delegates to scikit-image's `structural_similarity(img1, img2, channel_axis=0,
    data_range=img1.max() - img1.min())` (scikit-image is a dependency of the reference that is absent from this
    image, so this is a restatement of its published algorithm with the defaults that call selects — Wang et al.
    2004 with a 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, `reflect` borders, the border of
    (win-1)/2 pixels cropped before averaging, per-channel means averaged).  Parity unpinned: there is no
    scikit-image here to check it against; tests/test_container_abi.py checks its defining properties only.
    """
    from scipy.ndimage import uniform_filter

    a = img1.detach().cpu().numpy()
    b = img2.detach().cpu().numpy()
    if a.shape != b.shape or a.ndim != 3:
        raise ValueError("Input images must have the same (C, H, W) shape.")
    win = 7
    if min(a.shape[1:]) < win:
        raise ValueError("win_size exceeds image extent.")
    data_range = float(a.max() - a.min())
    a = a.astype(np.float64, copy=False)
    b = b.astype(np.float64, copy=False)
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    pad = (win - 1) // 2
    per_channel = []
    for x, y in zip(a, b):
        ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
        uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        per_channel.append(s[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return torch.tensor(np.mean(per_channel))
