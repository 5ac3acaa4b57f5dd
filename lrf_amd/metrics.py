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
