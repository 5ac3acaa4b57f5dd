"""svd_encode / svd_decode with the reference's signatures (lrf/compression/svd.py:117-361): the RGB branch — 8x8 patches
(the default, fused on the device), any other patch size, patch=False, uint8-quantised or float32 factors — on the MI355X.
The byte container stays on the host.

The YCbCr branch is not built: in the reference it raises TypeError for an integer `rank` (svd.py:234, 267) and the streams it
writes do not decode ("padded size" is appended twice per plane, svd.py:226, 237, so svd_decode reads the wrong entry)."""
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
    if color_space != "RGB":
        raise NotImplementedError("svd_encode(color_space='YCbCr') is defective in the reference (TypeError for an integer rank, "
                                  "streams that do not decode) and is not built")
    if image.dtype != torch.uint8:
        raise NotImplementedError("HIP path takes uint8 images")
    if dtype not in (torch.uint8, torch.float32):
        raise NotImplementedError("HIP path stores uint8-quantised or float32 factors")
    ps = tuple(patch_size) if patch else None
    H, W = image.shape[-2:]
    Hp, Wp, M, N = _lib.rgbspace_dims_any(H, W, ps)
    if rank is None:
        assert quality >= 0 and quality <= 100, "'quality' must be between 0 and 100."
        R = max(round(min(M, N) * quality / 100), 1)  # lrf/compression/svd.py:173-177
    else:
        R = rank
    ctx = _lib.context(image.device.index if image.is_cuda else None)
    dev = (image if image.is_cuda else image.cuda(ctx.device)).unsqueeze(0)
    nmat = 1 if ps is not None else 3
    sign = None
    if init_sign is not None:
        sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, R).expand(nmat, R).contiguous().cuda(ctx.device)
    metadata = {"dtype": str(image.dtype).split(".")[-1], "color space": color_space, "patch": patch}
    if ps is not None:
        metadata.update({"patch size": patch_size, "original size": [H, W], "padded size": [Hp, Wp]})
    if ps == (8, 8) and dtype is torch.uint8:  # the default branch, fused on the device
        U, V, qp = ctx.svd_encode_rgb(dev, R, sign)
        qp = qp[0].cpu().numpy()
        metadata["quantization"] = {"u": [float(qp[0]), float(qp[1])], "v": [float(qp[2]), float(qp[3])]}
        factors = [U[0].cpu().numpy(), V[0].cpu().numpy()]
    else:
        X = ctx.rgbspace_matrix_any(dev, ps)
        X = X[0] if ps is None else X  # [3, H, W]: three matrices; [1, M, N]: one
        u, v = ctx.svd_init(X, R, sign)  # u = U sqrt(s), v = (sqrt(s) Vh)^T (svd.py:179-183)
        if ps is not None:
            u, v = u[0], v[0]
        if dtype is torch.uint8:  # each factor tensor quantised as a whole (svd.py:185-187)
            qu, pu = ctx.quantize_u8(u.unsqueeze(0))
            qv, pv = ctx.quantize_u8(v.unsqueeze(0))
            pu, pv = pu[0].cpu().numpy(), pv[0].cpu().numpy()
            metadata["quantization"] = {"u": [float(pu[0]), float(pu[1])], "v": [float(pv[0]), float(pv[1])]}
            factors = [qu[0].cpu().numpy(), qv[0].cpu().numpy()]
        else:
            metadata["quantization"] = {"u": None, "v": None}
            factors = [u.cpu().numpy(), v.cpu().numpy()]
    return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in factors])])


def svd_decode(encoded_image: bytes) -> torch.Tensor:
    encoded_metadata, encoded_factors = separate_bytes(encoded_image, 2)
    metadata = bytes_to_dict(encoded_metadata)
    if metadata["color space"] != "RGB":
        raise NotImplementedError("svd_decode covers the RGB branch (the reference's YCbCr streams do not decode there either)")
    if metadata["dtype"] != "uint8":
        raise NotImplementedError("HIP decode writes uint8 images")
    u, v = [decode_tensor(f) for f in separate_bytes(encoded_factors, 2)]
    q = metadata["quantization"]
    quantised = q["u"] is not None and q["v"] is not None
    if (q["u"] is None) != (q["v"] is None) or (quantised and (u.dtype != np.uint8 or v.dtype != np.uint8)) or \
            (not quantised and (u.dtype != np.float32 or v.dtype != np.float32)):
        raise NotImplementedError("HIP decode takes uint8-quantised or float32 factors")
    R = int(u.shape[-1])
    if metadata["patch"]:
        ps = tuple(metadata["patch size"])
        H, W = metadata["original size"]
        Hp, Wp, M, N = _lib.rgbspace_dims_any(H, W, ps)
        if list(metadata["padded size"]) != [Hp, Wp] or tuple(u.shape) != (M, R) or tuple(v.shape) != (N, R):
            raise ValueError("stream factors do not match the reflect-padded patch geometry its metadata describes")
    else:
        ps = None
        if u.ndim != 3 or v.ndim != 3 or u.shape[0] != 3 or v.shape[0] != 3 or v.shape[2] != R:
            raise ValueError("stream factors are not [3, H, R] / [3, W, R]")
        H, W = int(u.shape[1]), int(v.shape[1])
    ctx = _lib.context(None)
    U = torch.from_numpy(np.array(u)).unsqueeze(0).cuda(ctx.device)
    V = torch.from_numpy(np.array(v)).unsqueeze(0).cuda(ctx.device)
    qp6 = None
    if quantised:  # dequantize subtracts the smallest stored code of the tensor (utils.py:241)
        qp6 = torch.tensor([[q["u"][0], q["u"][1], float(u.min()), q["v"][0], q["v"][1], float(v.min())]], dtype=torch.float32).cuda(ctx.device)
    if ps == (8, 8) and quantised:
        return ctx.svd_decode_rgb(U, V, qp6, H, W)[0].cpu()
    return ctx.svd_decode_any(U, V, H, W, ps, qp6)[0].cpu()
