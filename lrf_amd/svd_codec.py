"""svd_encode / svd_decode with the reference's signatures (lrf/compression/svd.py:117-361), default branch
(color_space="RGB", 8x8 patches, uint8-quantised factors) on the MI355X.  The byte container stays on the host."""
from typing import Optional

import numpy as np
import torch

from . import _lib
from .container import bytes_to_dict, combine_bytes, decode_tensor, dict_to_bytes, encode_tensor, separate_bytes


def svd_encode(image: torch.Tensor, rank: Optional[int] = None, quality=None, color_space: str = "RGB",
               scale_factor=(0.5, 0.5), patch: bool = True, patch_size=(8, 8), dtype: torch.dtype = None, **kwargs) -> bytes:
    assert (rank, quality) != (None, None), "Either 'rank' or 'quality' must be specified."
    init_sign = kwargs.pop("init_sign", None)
    if kwargs:
        raise TypeError(f"svd_encode() got unexpected keyword arguments {sorted(kwargs)}")
    dtype = image.dtype if dtype is None else dtype
    if color_space != "RGB" or not patch or tuple(patch_size) != (8, 8):
        raise NotImplementedError("HIP path covers svd_encode(color_space='RGB', patch=True, patch_size=(8,8))")
    if image.dtype != torch.uint8 or dtype is not torch.uint8:
        raise NotImplementedError("HIP path takes uint8 images and stores uint8-quantised factors")
    H, W = image.shape[-2:]
    Hp, Wp = H + (8 - H % 8) % 8, W + (8 - W % 8) % 8
    M = (Hp // 8) * (Wp // 8)
    if rank is None:
        assert quality >= 0 and quality <= 100, "'quality' must be between 0 and 100."
        R = max(round(min(M, 192) * quality / 100), 1)  # lrf/compression/svd.py:173-177
    else:
        R = rank
    ctx = _lib.context(image.device.index if image.is_cuda else None)
    dev = (image if image.is_cuda else image.cuda(ctx.device)).unsqueeze(0)
    sign = None
    if init_sign is not None:
        sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(1, R).contiguous().cuda(ctx.device)
    U, V, qp = ctx.svd_encode_rgb(dev, R, sign)
    qp = qp[0].cpu().numpy()
    metadata = {
        "dtype": str(image.dtype).split(".")[-1],
        "color space": color_space,
        "patch": patch,
        "patch size": patch_size,
        "original size": [H, W],
        "padded size": [Hp, Wp],
        "quantization": {"u": [float(qp[0]), float(qp[1])], "v": [float(qp[2]), float(qp[3])]},
    }
    factors = [U[0].cpu().numpy(), V[0].cpu().numpy()]
    return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(f) for f in factors])])


def svd_decode(encoded_image: bytes) -> torch.Tensor:
    encoded_metadata, encoded_factors = separate_bytes(encoded_image, 2)
    metadata = bytes_to_dict(encoded_metadata)
    if metadata["color space"] != "RGB" or not metadata["patch"] or list(metadata["patch size"]) != [8, 8]:
        raise NotImplementedError("HIP decode covers the RGB / 8x8-patch branch of svd_decode")
    u, v = [decode_tensor(f) for f in separate_bytes(encoded_factors, 2)]
    q = metadata["quantization"]
    if q["u"] is None or q["v"] is None or u.dtype != np.uint8 or metadata["dtype"] != "uint8":
        raise NotImplementedError("HIP decode takes uint8-quantised factors")
    H, W = metadata["original size"]
    ctx = _lib.context(None)
    qp6 = torch.tensor([[q["u"][0], q["u"][1], float(u.min()), q["v"][0], q["v"][1], float(v.min())]], dtype=torch.float32)
    rgb = ctx.svd_decode_rgb(torch.from_numpy(np.ascontiguousarray(u)).cuda(ctx.device).unsqueeze(0),
                             torch.from_numpy(np.ascontiguousarray(v)).cuda(ctx.device).unsqueeze(0), qp6.cuda(ctx.device), H, W)
    return rgb[0].cpu()
