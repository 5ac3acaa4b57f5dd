"""qmf_encode / qmf_decode with the reference's signatures (lrf/compression/qmf.py:116-353), running the
arithmetic on the MI355X.  Python keeps the byte container (JSON metadata + per-column zlib), exactly
as the reference does on the host."""
import math
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .container import (bytes_to_dict, combine_bytes, decode_tensor, dict_to_bytes, encode_tensor, separate_bytes)


def qmf_ranks(image_hw, rank=None, quality=None):
    """The (Y, Cb, Cr) ranks qmf_encode uses: lrf/compression/qmf.py:215-225, 244-250."""
    H, W = image_hw
    if not isinstance(rank, Iterable):
        rank = (None, None, None) if rank is None else (rank, max(rank // 2, 1), max(rank // 2, 1))
    if not isinstance(quality, Iterable):
        quality = (None, None, None) if quality is None else (quality, quality / 2, quality / 2)
    out = []
    for i, (_, _, _, _, M) in enumerate(_lib.plane_dims(H, W)):
        if rank[i] is None:
            assert quality[i] >= 0 and quality[i] <= 100, "'quality' must be between 0 and 100."
            out.append(max(round(min(M, 64) * quality[i] / 100), 1))
        else:
            out.append(rank[i])
    return out


def _check_hip_branch(color_space, scale_factor, patch, patch_size, bounds, dtype, kwargs):
    """Splits qmf_encode's **kwargs (lrf/compression/qmf.py:127, forwarded to QMF(...) at :189, 208, 256, 280) into the loop
    parameters of the tuned path and the options only the general solver takes (`qmf_opts`: l2, l1_ratio, eps, num_levels)."""
    if color_space == "YCbCr" and not (len(scale_factor) == 2 and min(scale_factor) > 0):
        raise ValueError("scale_factor must be two positive numbers")
    if dtype is not torch.int8:
        raise NotImplementedError("HIP path stores int8 factors only")
    num_iters = kwargs.pop("num_iters", 10)
    init_sign = kwargs.pop("init_sign", None)
    kwargs.pop("init", None)  # explicit (u0, v0) fp32 initial factors (tests)
    kwargs.pop("verbose", None)
    for dup in ("factor", "project"):  # the reference passes these itself: a caller's copy is a duplicate keyword there too
        if dup in kwargs:
            raise TypeError(f"QMF() got multiple values for keyword argument '{dup}'")
    qmf_opts = {}
    l2 = kwargs.pop("l2", 0)
    l2p = tuple(l2) if isinstance(l2, (tuple, list)) else (l2, l2)
    if any(v != 0 for v in l2p):
        qmf_opts["l2"] = l2
        qmf_opts["l1_ratio"] = kwargs.pop("l1_ratio", 0)
    else:
        kwargs.pop("l1_ratio", None)  # without an l2 weight the l1 share multiplies zero (qmf.py:154-157)
    eps = kwargs.pop("eps", 1e-16)
    if eps != 1e-16:
        qmf_opts["eps"] = eps
    num_levels = kwargs.pop("num_levels", None)
    if num_levels:
        qmf_opts["num_levels"] = num_levels
    if kwargs:
        raise TypeError(f"CoordinateDescent.__init__() got an unexpected keyword argument '{sorted(kwargs)[0]}'")
    if num_iters < 0:
        raise ValueError("num_iters must be >= 0")
    lo, hi = math.ceil(bounds[0]), math.floor(bounds[1])
    return num_iters, lo, hi, init_sign, qmf_opts


def qmf_factorize_batch(images: torch.Tensor, ranks: Sequence[int], num_iters: int = 10, bounds=(-16, 15),
                        init_sign=None, out=None):
    """GPU-only part of the encoder for a batch [B,3,H,W] of uint8 images already in HBM.

    Returns (U, V): int8 CUDA tensors [B, sum_c M_c R_c] and [B, 64 sum_c R_c] holding, per image, the factors of
    the Y, Cb, Cr planes back to back (row-major [M_c, R_c] / [64, R_c])."""
    assert images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[1] == 3
    ctx = _lib.context(images.device.index)
    sign = None
    if init_sign is not None:
        sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, sum(ranks))
        sign = sign.expand(images.shape[0], sum(ranks)).contiguous().cuda(images.device)
    return ctx.encode_rgb(images.contiguous(), list(ranks), num_iters, math.ceil(bounds[0]), math.floor(bounds[1]),
                          sign, out=out)


def qmf_encode_sweep(images: torch.Tensor, qualities=None, ranks=None, bounds=(-16, 15), num_iters: int = 10, init_sign=None,
                     pack_workers: Optional[int] = None) -> list:
    """qmf_encode of a batch [B,3,H,W] at every quality of `qualities` (or every rank / rank triple of `ranks`) in ONE GPU
    call (lrf_qmf_encode_sweep_rgb_u8: the R-D sweep of experiments/comparison/eval.py:83-110; default branch — YCbCr, 8x8
    patches).  Returns one list of B byte streams per quality, each stream byte-identical to `qmf_encode(image, quality=q)`.
    Qualities that give the same rank triple are computed once; rank triples above 32 fall back to one call each."""
    assert (qualities is None) != (ranks is None), "give either qualities or ranks"
    H, W = images.shape[-2:]
    params = list(qualities) if qualities is not None else list(ranks)
    triples = [tuple(qmf_ranks((H, W), None, p)) if qualities is not None else tuple(qmf_ranks((H, W), p, None)) for p in params]
    if images.dtype != torch.uint8:
        raise NotImplementedError("HIP path takes uint8 images")
    ctx = _lib.context(images.device.index if images.is_cuda else None)
    dev = (images if images.is_cuda else images.cuda(ctx.device)).contiguous()
    B = dev.shape[0]
    lo, hi = math.ceil(bounds[0]), math.floor(bounds[1])
    unique = sorted(set(triples))
    fused = [t for t in unique if max(t) <= 32]
    factors = {}
    if fused and num_iters >= 1:
        rmax = [max(t[c] for t in fused) for c in range(3)]
        sign = None
        if init_sign is not None:  # [Rmax_Y + Rmax_Cb + Rmax_Cr] or [B, ...]: the signs of the largest ranks' components
            sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, sum(rmax)).expand(B, sum(rmax)).contiguous().cuda(dev.device)
        for t, (U, V) in zip(fused, ctx.encode_sweep_rgb(dev, fused, num_iters, lo, hi, sign)):
            factors[t] = (U, V)
    out = {}
    for t in unique:
        if t in factors:
            Uh, Vh = (x.numpy() for x in ctx.to_host(*factors[t]))
            out[t] = pack_streams_native(Uh, Vh, (H, W), list(t), bounds, (8, 8), "uint8", threads=pack_workers or default_pack_threads())
        else:  # ranks above 32 (or num_iters = 0): the per-triple encoder
            if init_sign is not None:
                raise NotImplementedError("init_sign with rank triples outside the fused sweep")
            out[t] = qmf_encode_batch(dev, rank=list(t), bounds=bounds, num_iters=num_iters, pack_workers=pack_workers)
    return [out[t] for t in triples]


def qmf_factorize_host(images: torch.Tensor, ranks: Sequence[int], num_iters: int = 10, bounds=(-16, 15), init_sign=None,
                       out=None, slots: int = 2, sub_batch: int = 0, device=None):
    """Host -> host form of qmf_factorize_batch: `images` is a uint8 CPU tensor [B,3,H,W]
    (page-locked — torch's pin_memory — for link-speed copies), the int8 factors come back as CPU tensors.  The batch
    streams through the pipelined encoder (include/lrf_hip.h, lrf_pipe): uploads, kernels and downloads of different
    sub-batches overlap.  Same values as qmf_factorize_batch, bit for bit."""
    release_plane_lanes()  # idle contexts / streams of small any-shape calls would cost the pipe's schedule
    pipe = _lib.pipe(device, slots, sub_batch)
    return pipe.encode_rgb_host(images, list(ranks), num_iters, math.ceil(bounds[0]), math.floor(bounds[1]), init_sign, out)


def split_factors(U_row: np.ndarray, V_row: np.ndarray, image_hw, ranks):
    """One image's packed factor rows -> [u_y, v_y, u_cb, v_cb, u_cr, v_cr] numpy int8 matrices."""
    H, W = image_hw
    out, uo, vo = [], 0, 0
    for (_, _, _, _, M), R in zip(_lib.plane_dims(H, W), ranks):
        out.append(U_row[uo:uo + M * R].reshape(M, R))
        out.append(V_row[vo:vo + 64 * R].reshape(64, R))
        uo += M * R
        vo += 64 * R
    return out


def pack_image(factors, image_hw, ranks, bounds, patch_size=(8, 8), dtype_name="uint8") -> bytes:
    """metadata + six factor blobs -> the reference's byte stream (lrf/compression/qmf.py:157-162,233-254,288-290)."""
    H, W = image_hw
    dims = _lib.plane_dims(H, W)
    metadata = {
        "dtype": dtype_name,
        "color space": "YCbCr",
        "patch": True,
        "bounds": bounds,
        "patch size": patch_size,
        "original size": [[d[0], d[1]] for d in dims],
        "padded size": [[d[2], d[3]] for d in dims],
        "rank": list(ranks),
    }
    return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(f) for f in factors])])


_PACK_POOL = None
_PACK_LIB = None


def default_pack_threads() -> int:
    """Host threads for the zlib-9 container packing when the caller names none: the CPUs this process may actually use —
    the smaller of its affinity mask and its cgroup CPU quota (a GPU box here: 256 CPUs in the mask, a quota of 16; more
    threads than the quota borrow against it and are then throttled, DESIGN.md "bytes out")."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1") and int(period) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return max(1, min(n, 64))


def _pack_lib():
    """liblrf_pack.so (include/lrf_pack.h): the same container built by native host threads."""
    global _PACK_LIB
    if _PACK_LIB is None:
        import ctypes
        import os
        # LRF_PACK_LIB: another build of lrf_pack.cpp (the sanitizer builds of tools/run_sanitizers.sh)
        path = os.environ.get("LRF_PACK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblrf_pack.so")
        lib = ctypes.CDLL(path)
        lib.lrf_pack_qmf_streams.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p,
                                             ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_int64)]
        lib.lrf_pack_qmf_streams_planes.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int),
                                                    ctypes.c_int64, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int,
                                                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
        lib.lrf_pack_free.argtypes = [ctypes.c_void_p]
        lib.lrf_pack_free.restype = None
        lib.lrf_pack_unpack_qmf_factors.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64,
                                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        # Byte identity with the reference (CPython's zlib module at level 9) needs the same deflate implementation:
        # a Python built against another zlib (conda, zlib-ng) would make the native streams valid but different.
        import zlib
        lib.lrf_pack_zlib_version.restype = ctypes.c_char_p
        native = (lib.lrf_pack_zlib_version() or b"").decode(errors="replace")
        if native != zlib.ZLIB_RUNTIME_VERSION:
            raise OSError(f"liblrf_pack.so links zlib {native}, this Python runs zlib {zlib.ZLIB_RUNTIME_VERSION}: "
                          "using the Python container code so that streams stay byte-identical to the reference's")
        _PACK_LIB = lib
    return _PACK_LIB


def pack_streams_native(Uh: np.ndarray, Vh: np.ndarray, image_hw, ranks, bounds, patch_size=(8, 8), dtype_name="uint8",
                        threads: int = 0) -> list:
    """All images of a batch -> byte streams through liblrf_pack.so (byte-identical to pack_image)."""
    import ctypes
    H, W = image_hw
    dims = _lib.plane_dims(H, W)
    metadata = dict_to_bytes({
        "dtype": dtype_name, "color space": "YCbCr", "patch": True, "bounds": bounds, "patch size": patch_size,
        "original size": [[d[0], d[1]] for d in dims], "padded size": [[d[2], d[3]] for d in dims], "rank": list(ranks)})
    Uh = np.ascontiguousarray(Uh, dtype=np.int8)
    Vh = np.ascontiguousarray(Vh, dtype=np.int8)
    B = Uh.shape[0]
    M = (ctypes.c_int64 * 3)(*[d[4] for d in dims])
    R = (ctypes.c_int * 3)(*[int(r) for r in ranks])
    out = (ctypes.c_void_p * B)()
    lens = (ctypes.c_int64 * B)()
    rc = _pack_lib().lrf_pack_qmf_streams(Uh.ctypes.data_as(ctypes.c_void_p), Uh.shape[1], Vh.ctypes.data_as(ctypes.c_void_p),
                                          Vh.shape[1], B, M, R, metadata, len(metadata), int(threads), out, lens)
    if rc:
        raise RuntimeError(f"lrf_pack_qmf_streams failed ({rc})")
    streams = []
    for b in range(B):
        streams.append(ctypes.string_at(out[b], lens[b]))
        _pack_lib().lrf_pack_free(out[b])
    return streams


def anyshape_metadata(image_hw, ranks, bounds, patch_size, dtype_name="uint8", chroma=None) -> dict:
    """Metadata of the patch-size / patch=False streams, keys in the reference's order (qmf.py:157-162, 233-254, 265-277)."""
    dims = _lib.plane_dims_any(image_hw[0], image_hw[1], patch_size, chroma)
    metadata = {"dtype": dtype_name, "color space": "YCbCr", "patch": patch_size is not None, "bounds": bounds}
    if patch_size is not None:
        metadata["patch size"] = patch_size
        metadata["original size"] = [[d[0], d[1]] for d in dims]
        metadata["padded size"] = [[d[2], d[3]] for d in dims]
    else:
        metadata["original size"] = [[d[0], d[1]] for d in dims]
    metadata["rank"] = list(ranks)
    return metadata


def pack_anyshape_native(per_plane, image_hw, ranks, bounds, patch_size, dtype_name="uint8", chroma=None, threads: int = 0) -> list:
    """All images of a batch of the patch-size / patch=False branches -> byte streams through liblrf_pack.so
    (lrf_pack_qmf_streams_planes; byte-identical to pack_anyshape per image, tests/test_container_abi.py).
    per_plane: three (u [B,M,R], v [B,N,R]) int8 numpy pairs."""
    import ctypes
    metadata = dict_to_bytes(anyshape_metadata(image_hw, ranks, bounds, patch_size, dtype_name, chroma))
    arrs = []
    for u, v in per_plane:
        arrs += [np.ascontiguousarray(u, dtype=np.int8), np.ascontiguousarray(v, dtype=np.int8)]
    B = arrs[0].shape[0]
    F = (ctypes.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    rows = (ctypes.c_int64 * 6)(*[a.shape[1] for a in arrs])
    cols = (ctypes.c_int * 6)(*[a.shape[2] for a in arrs])
    out = (ctypes.c_void_p * B)()
    lens = (ctypes.c_int64 * B)()
    rc = _pack_lib().lrf_pack_qmf_streams_planes(F, rows, cols, B, 0 if patch_size is not None else 1, metadata, len(metadata), int(threads),
                                                 out, lens)
    if rc:
        raise RuntimeError(f"lrf_pack_qmf_streams_planes failed ({rc})")
    streams = []
    for b in range(B):
        streams.append(ctypes.string_at(out[b], lens[b]))
        _pack_lib().lrf_pack_free(out[b])
    return streams


def _pack_pool(workers):
    """zlib releases the GIL, so the per-column zlib-9 packing of finished images scales over host threads
    (SURVEY.md §8f N2); the streams stay byte-identical to the serial ones."""
    global _PACK_POOL
    from concurrent.futures import ThreadPoolExecutor
    if _PACK_POOL is None or _PACK_POOL._max_workers != workers:
        _PACK_POOL = ThreadPoolExecutor(max_workers=workers)
    return _PACK_POOL


def qmf_encode_batch(images: torch.Tensor, rank=None, quality=None, bounds=(-16, 15), num_iters: int = 10,
                     init_sign=None, pack_workers: Optional[int] = None, patch: bool = True, patch_size=(8, 8)) -> list:
    """Batched qmf_encode (YCbCr branch) -> list of byte streams, one per image.  The factorisation of the whole batch
    runs on the GPU; for the default 8x8 patches the byte containers are packed by liblrf_pack.so on native host threads
    (`pack_workers` None / 0: as many as this process may use, default_pack_threads), or, with pack_workers="python", by the Python container code on a
    thread pool.  Other patch sizes and patch=False go through the any-shape kernels (container packed in Python)."""
    assert (rank, quality) != (None, None), "Either 'rank' or 'quality' must be specified."
    H, W = images.shape[-2:]
    if (not images.is_cuda) and patch and tuple(patch_size) == (8, 8) and images.dtype == torch.uint8 and num_iters >= 1 \
            and pack_workers != "python" and not (isinstance(pack_workers, int) and pack_workers < 0):
        # host tensor in, byte streams out: the pipelined encoder, the container of each finished sub-batch packed by
        # liblrf_pack.so while the GPU works on the next ones
        try:
            _pack_lib()
        except OSError:
            pass
        else:
            ranks = qmf_ranks((H, W), rank, quality)
            release_plane_lanes()  # idle contexts / streams of small any-shape calls would cost the pipe's schedule
            pipe = _lib.pipe(None)
            streams = []
            for first, n, U, V in pipe.encode_rgb_host_iter(images, ranks, num_iters, math.ceil(bounds[0]), math.floor(bounds[1]),
                                                            init_sign):
                streams += pack_streams_native(U[first:first + n].numpy(), V[first:first + n].numpy(), (H, W), ranks, bounds,
                                               (8, 8), "uint8", threads=pack_workers or default_pack_threads())
            return streams
    ctx = _lib.context(images.device.index if images.is_cuda else None)
    dev = images if images.is_cuda else images.cuda(ctx.device)
    if not patch or tuple(patch_size) != (8, 8):
        if images.dtype != torch.uint8:
            raise NotImplementedError("HIP path takes uint8 images")
        lo, hi = math.ceil(bounds[0]), math.floor(bounds[1])
        return _qmf_encode_anyshape(ctx, dev, rank, quality, bounds, (lo, hi), tuple(patch_size) if patch else None, num_iters,
                                    init_sign, None)
    ranks = qmf_ranks((H, W), rank, quality)
    U, V = qmf_factorize_batch(dev, ranks, num_iters, bounds, init_sign)
    Uh, Vh = (t.numpy() for t in ctx.to_host(U, V))  # waits for the stream, then raises if a launch of this call gave up
    dtype_name = str(images.dtype).split(".")[-1]
    if pack_workers != "python" and not (isinstance(pack_workers, int) and pack_workers < 0):
        try:
            return pack_streams_native(Uh, Vh, (H, W), ranks, bounds, (8, 8), dtype_name, threads=pack_workers or default_pack_threads())
        except OSError:  # liblrf_pack.so missing or linked against another zlib: the Python container code, same bytes
            pack_workers = "python"
    pack_workers = None if pack_workers == "python" else -pack_workers

    def pack(b):
        return pack_image(split_factors(Uh[b], Vh[b], (H, W), ranks), (H, W), ranks, bounds, (8, 8), dtype_name)

    import os
    workers = pack_workers if pack_workers is not None else min(32, os.cpu_count() or 1)
    if workers <= 1 or images.shape[0] == 1:
        return [pack(b) for b in range(images.shape[0])]
    return list(_pack_pool(workers).map(pack, range(images.shape[0])))


def qmf_encode(image: torch.Tensor, rank=None, quality=None, color_space: str = "YCbCr",
               scale_factor=(0.5, 0.5), patch: bool = True, patch_size=(8, 8), bounds=(-16, 15),
               dtype: torch.dtype = torch.int8, **kwargs) -> bytes:
    """QMF compression of one image [3,H,W]: same signature and byte stream as the reference's
    lrf.qmf_encode (lrf/compression/qmf.py:116-292)."""
    assert (rank, quality) != (None, None), "Either 'rank' or 'quality' must be specified."
    assert color_space in ("RGB", "YCbCr"), "`color_space` must be one of 'RGB' or 'YCbCr'."
    num_iters, lo, hi, init_sign, qmf_opts = _check_hip_branch(color_space, scale_factor, patch, patch_size, bounds, dtype,
                                                               dict(kwargs))
    if image.dtype != torch.uint8:
        raise NotImplementedError("HIP path takes uint8 images")
    H, W = image.shape[-2:]
    ctx = _lib.context(image.device.index if image.is_cuda else None)
    dev = (image if image.is_cuda else image.cuda(ctx.device)).unsqueeze(0)
    if qmf_opts:  # l2 / l1_ratio / eps / num_levels: the general solver behind the same driver (qmf.py:256 forwards them to QMF)
        return _qmf_encode_general(ctx, dev, color_space, rank, quality, bounds, tuple(patch_size) if patch else None, num_iters,
                                   init_sign, kwargs.get("init"), _lib.chroma_size(H, W, scale_factor) if color_space == "YCbCr" else None,
                                   qmf_opts)
    if color_space == "RGB":
        return _qmf_encode_rgbspace(ctx, dev, rank, quality, bounds, (lo, hi), tuple(patch_size) if patch else None, num_iters, init_sign,
                                    kwargs.get("init"))
    chroma = _lib.chroma_size(H, W, scale_factor)
    if not patch or tuple(patch_size) != (8, 8) or chroma is not None:  # the any-shape path (also 8x8 with another scale factor)
        return _qmf_encode_anyshape(ctx, dev, rank, quality, bounds, (lo, hi), tuple(patch_size) if patch else None, num_iters,
                                    init_sign, kwargs.get("init"), chroma)[0]
    ranks = qmf_ranks((H, W), rank, quality)
    if num_iters == 0:
        factors = _svd_init_factors(ctx, dev, ranks, init_sign)
    else:
        U, V = qmf_factorize_batch(dev, ranks, num_iters, (lo, hi), init_sign)
        Uh, Vh = (t.numpy() for t in ctx.to_host(U, V))
        try:  # the container through liblrf_pack.so: the columns of the six factors are zlib-packed on host threads
            return pack_streams_native(Uh, Vh, (H, W), ranks, bounds, patch_size, str(image.dtype).split(".")[-1],
                                       threads=default_pack_threads())[0]
        except OSError:  # library not built: the same bytes from the Python container code
            factors = split_factors(Uh[0], Vh[0], (H, W), ranks)
    return pack_image(factors, (H, W), ranks, bounds, patch_size, str(image.dtype).split(".")[-1])


def anyshape_ranks(image_hw, patch_size, rank=None, quality=None, chroma=None):
    """Ranks of (Y, Cb, Cr) for patches `patch_size` (lrf/compression/qmf.py:244-250) or, with patch_size None,
    for patch=False (:268-274): max(round(min(M, N) * quality / 100), 1) on the matrix each plane becomes."""
    H, W = image_hw
    if not isinstance(rank, Iterable):
        rank = (None, None, None) if rank is None else (rank, max(rank // 2, 1), max(rank // 2, 1))
    if not isinstance(quality, Iterable):
        quality = (None, None, None) if quality is None else (quality, quality / 2, quality / 2)
    out = []
    for i, d in enumerate(_lib.plane_dims_any(H, W, patch_size, chroma)):
        if rank[i] is None:
            assert quality[i] >= 0 and quality[i] <= 100, "'quality' must be between 0 and 100."
            out.append(max(round(min(d[4], d[5]) * quality[i] / 100), 1))
        else:
            out.append(rank[i])
    return out


def _qmf_encode_anyshape(ctx, dev, rank, quality, bounds, int_bounds, patch_size, num_iters, init_sign, init, chroma=None):
    """qmf_encode(color_space="YCbCr") with a patch size other than 8x8 (lrf/compression/qmf.py:232-262) or with
    patch=False (patch_size None, :264-286) for a batch dev [B,3,H,W]: per plane one matrix [M, N] per image, all images
    of a plane factorised in one call of the any-shape kernels.  Returns one byte stream per image.
    `init`: optional three (u0, v0) fp32 pairs ([M,R] / [N,R], or with a leading batch axis) replacing the SVD
    initialisation (tests); `init_sign`: [R0+R1+R2] or [B, R0+R1+R2]."""
    B = dev.shape[0]
    H, W = dev.shape[-2:]
    dims = _lib.plane_dims_any(H, W, patch_size, chroma)
    ranks = anyshape_ranks((H, W), patch_size, rank, quality, chroma)
    # Small calls: the three planes on three contexts / streams.  These kernels give a matrix to one workgroup, so one image
    # keeps a CU or two busy per plane and its latency is the SUM of the planes' chains unless they run side by side (one
    # 512x768 image without patches: 57 -> 45 ms); a large batch fills the chip by itself (three streams: no gain).
    side_by_side = B <= 8 and torch.cuda.is_available()
    offs = [sum(ranks[:c]) for c in range(3)]

    def one_plane(pctx, c):
        X = pctx.planes_any(dev, patch_size, c, chroma)
        R = ranks[c]
        sign = None
        if init_sign is not None:
            sg = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, sum(ranks))[:, offs[c]:offs[c] + R]
            sign = sg.expand(B, R).contiguous().cuda(dev.device)
        if init is not None:
            u0 = torch.as_tensor(init[c][0], dtype=torch.float32).reshape(-1, dims[c][4], R).expand(B, -1, -1).contiguous().cuda(dev.device)
            v0 = torch.as_tensor(init[c][1], dtype=torch.float32).reshape(-1, dims[c][5], R).expand(B, -1, -1).contiguous().cuda(dev.device)
            u, v = (u0.cpu().to(torch.int8), v0.cpu().to(torch.int8)) if num_iters == 0 else \
                pctx.bcd(X, u0, v0, num_iters, int_bounds[0], int_bounds[1])
        elif num_iters == 0:  # the float factors go straight through torch's truncating cast (qmf.py:258-260)
            u0, v0 = pctx.svd_init(X, R, sign)
            u, v = u0.cpu().to(torch.int8), v0.cpu().to(torch.int8)
        else:
            u, v = pctx.decompose(X, R, num_iters, int_bounds[0], int_bounds[1], sign)
        return u, v, X  # X: kept alive until its stream has been waited for

    if side_by_side:
        ctxs, lanes = _plane_lanes(dev.device)
        cur = torch.cuda.current_stream(dev.device)
        pending = []
        for c in range(3):
            lanes[c].wait_stream(cur)
            with torch.cuda.stream(lanes[c]):
                pending.append(one_plane(ctxs[c], c))
        for c in range(3):
            cur.wait_stream(lanes[c])
        per_plane = [(u.cpu().numpy(), v.cpu().numpy()) for u, v, _ in pending]  # [B, M, R], [B, N, R]
    else:
        per_plane = []
        for c in range(3):
            u, v, _ = one_plane(ctx, c)
            per_plane.append((u.cpu().numpy(), v.cpu().numpy()))
    dtype_name = str(dev.dtype).split(".")[-1]
    try:  # the containers of the whole batch on native host threads (a column, or a whole factor, per work item)
        return pack_anyshape_native(per_plane, (H, W), ranks, bounds, patch_size, dtype_name, chroma)
    except OSError:  # liblrf_pack.so absent or linked against another zlib: the Python container code, same bytes
        pass
    streams = []
    for b in range(B):
        factors = []
        for u, v in per_plane:
            # patch=False keeps the plane's channel axis: the factors are 3-D there (qmf.py:281-282), 2-D with patches
            factors += [u[b:b + 1], v[b:b + 1]] if patch_size is None else [u[b], v[b]]
        streams.append(pack_anyshape(factors, (H, W), ranks, bounds, patch_size, dtype_name, chroma))
    return streams


def _qmf_encode_general(ctx, dev, color_space, rank, quality, bounds, patch_size, num_iters, init_sign, init, chroma, qmf_opts):
    """qmf_encode with QMF options beyond the tuned configuration — `l2`, `l1_ratio`, `eps`, `num_levels`, which the
    reference forwards to QMF(rank, bounds, factor=(0, 1), **kwargs) per matrix (lrf/compression/qmf.py:189, 208, 256, 280) —
    for one image dev [1,3,H,W]: the same matrices as the other branches (any patch size, patch=False, both colour spaces),
    each through lrf_amd.QMF's general path (lrf_qmf_decompose_ex_f32), the float factors through torch's cast to int8
    (qmf.py:258-260) and the same container.  `init`: per matrix an (u0, v0) pair replacing the SVD initialisation (tests)."""
    from .factorization import QMF
    H, W = dev.shape[-2:]
    if color_space == "RGB":
        if isinstance(rank, (list, tuple)) or isinstance(quality, (list, tuple)):
            raise ValueError("color_space='RGB' takes a scalar rank / quality")
        Hp, Wp, M, N = _lib.rgbspace_dims_any(H, W, patch_size)
        if rank is None:
            assert quality >= 0 and quality <= 100, "'quality' must be between 0 and 100."
            R = max(round(min(M, N) * quality / 100), 1)
        else:
            R = rank
        X = ctx.rgbspace_matrix_any(dev, patch_size)
        mats = [X[0]] if patch_size is None else [X]  # [3, H, W]: three matrices in one call; [1, M, N]: one
        ranks = [R]
    else:
        ranks = anyshape_ranks((H, W), patch_size, rank, quality, chroma)
        mats = [ctx.planes_any(dev, patch_size, c, chroma) for c in range(3)]
    offs = [sum(ranks[:c]) for c in range(len(ranks))]
    factors = []
    for c, (Xc, R) in enumerate(zip(mats, ranks)):
        nb = Xc.shape[0]
        sign = None
        if init_sign is not None:
            sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, sum(ranks))[:, offs[c]:offs[c] + R].expand(nb, R).contiguous()
        opts = dict(qmf_opts)
        num_levels = opts.pop("num_levels", None)
        if init is not None:  # the reference's initial factors (fixtures): straight into the general loop
            u0 = torch.as_tensor(init[c][0], dtype=torch.float32).reshape(nb, Xc.shape[1], R)
            v0 = torch.as_tensor(init[c][1], dtype=torch.float32).reshape(nb, Xc.shape[2], R)
            w0 = None if len(init[c]) < 3 else torch.as_tensor(init[c][2], dtype=torch.float32).reshape(1, 2).expand(nb, 2)
            if num_iters == 0:
                u, v = u0, v0
            else:
                u, v, _ = ctx.decompose_ex(Xc, R, num_iters, bounds, opts.get("l2", 0), opts.get("l1_ratio", 0), (0, 1), None,
                                           init=(u0, v0), eps=opts.get("eps", 1e-16), w_init=w0)
        else:
            u, v, _ = QMF(rank=R, num_iters=num_iters, bounds=bounds, num_levels=num_levels, factor=(0, 1), init_sign=sign,
                          **opts).decompose(Xc)
        u8, v8 = u.cpu().to(torch.int8).numpy(), v.cpu().to(torch.int8).numpy()  # the affine pair w is dropped, as at qmf.py:257
        if color_space == "RGB":
            factors += [u8[0], v8[0]] if patch_size is not None else [u8, v8]
        else:
            factors += [u8[0:1], v8[0:1]] if patch_size is None else [u8[0], v8[0]]
    dtype_name = str(dev.dtype).split(".")[-1]
    if color_space == "RGB":
        metadata = {"dtype": dtype_name, "color space": "RGB", "patch": patch_size is not None, "bounds": bounds}
        if patch_size is not None:
            metadata.update({"patch size": patch_size, "original size": [H, W], "padded size": [Hp, Wp], "rank": ranks[0]})
        else:
            metadata["rank"] = ranks[0]
        return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in factors])])
    return pack_anyshape(factors, (H, W), ranks, bounds, patch_size, dtype_name, chroma)


_PLANE_LANES = {}
_PLANE_LANES_LOCK = __import__("threading").Lock()


def _plane_lanes(device):
    """three contexts and three streams per (host thread, device) for the plane-parallel small calls of _qmf_encode_anyshape.
    They are released again (release_plane_lanes) before this thread's pipelined batch encoder runs: every extra context /
    stream alive in the process costs the pipe's schedule (one unused context and two streams moved
    256 x 512x768 host -> host from 6.1 to 7-9 ms), and a lane's workspace is a few hundred MB after a large patch=False call."""
    import threading
    key = (threading.get_ident(), torch.device(device).index or 0)
    with _PLANE_LANES_LOCK:
        lanes = _PLANE_LANES.get(key)
    if lanes is None:
        lanes = ([_lib.Context(key[1]) for _ in range(3)], [torch.cuda.Stream(device=key[1]) for _ in range(3)])
        with _PLANE_LANES_LOCK:
            _PLANE_LANES[key] = lanes
    return lanes


def release_plane_lanes(all_threads: bool = False) -> int:
    """Destroys the plane lanes of the calling thread: their contexts (workspaces, streams) and torch streams.  Called by
    qmf_encode_batch's host path; safe to call any time (the lanes are re-created on demand).  all_threads=True also takes
    the lanes of OTHER threads — only for a quiescent process (tests, shutdown): a lane another thread is encoding on would
    be closed under it.  Returns the number of lane sets released."""
    import threading
    me = threading.get_ident()
    n = 0
    with _PLANE_LANES_LOCK:
        mine = [(k, _PLANE_LANES.pop(k)) for k in list(_PLANE_LANES) if all_threads or k[0] == me]
    for key, (ctxs, lanes) in mine:
        for st in lanes:
            st.synchronize()
        for c in ctxs:
            c.close()
        n += 1
    return n


def pack_anyshape(factors, image_hw, ranks, bounds, patch_size, dtype_name="uint8", chroma=None) -> bytes:
    """Byte stream of the patch-size / patch=False branches (metadata keys in the reference's order, qmf.py:157-162,
    233-254, 265-277, 288-290).  factors: [u_y, v_y, u_cb, v_cb, u_cr, v_cr] int8, 2-D with patches, [1, rows, R]
    without (the reference keeps the plane's channel axis there and encode_tensor stores such tensors whole)."""
    metadata = anyshape_metadata(image_hw, ranks, bounds, patch_size, dtype_name, chroma)
    return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in factors])])


def _qmf_decode_anyshape(encoded_image: bytes, device=None) -> torch.Tensor:
    """YCbCr branch of qmf_decode for any patch size / patch=False (qmf.py:325-351) -> uint8 CUDA tensor [3,H,W]."""
    encoded_metadata, encoded_factors = separate_bytes(encoded_image, 2)
    metadata = bytes_to_dict(encoded_metadata)
    if metadata["dtype"] != "uint8":
        raise NotImplementedError("HIP decode writes uint8 images")
    patch_size = tuple(metadata["patch size"]) if metadata["patch"] else None
    f = [decode_tensor(x) for x in separate_bytes(encoded_factors, 6)]
    H, W = metadata["original size"][0]
    chroma = tuple(int(v) for v in metadata["original size"][1])  # any scale_factor: the stream carries the plane sizes
    if chroma == (H // 2, W // 2):
        chroma = None
    dims = _lib.plane_dims_any(H, W, patch_size, chroma)
    for c in range(3):
        if list(metadata["original size"][c]) != [dims[c][0], dims[c][1]] or \
                (patch_size is not None and list(metadata["padded size"][c]) != [dims[c][2], dims[c][3]]):
            raise ValueError("stream geometry is not the reflect-padded layout its plane sizes imply (Cb and Cr must agree)")
        if tuple(f[2 * c].shape[-2:]) != (dims[c][4], f[2 * c].shape[-1]) or tuple(f[2 * c + 1].shape[-2:]) != (dims[c][5], f[2 * c].shape[-1]):
            raise ValueError("stream factors do not match the plane geometry its metadata describes")
    ctx = _lib.context(device)
    Us, Vs = [], []
    for c in range(3):
        u = np.array(f[2 * c], dtype=np.int8)  # copies: decode_tensor may hand back read-only views
        v = np.array(f[2 * c + 1], dtype=np.int8)
        Us.append(torch.from_numpy(u.reshape(1, dims[c][4], -1)).cuda(ctx.device))
        Vs.append(torch.from_numpy(v.reshape(1, dims[c][5], -1)).cuda(ctx.device))
    return ctx.decode_any(Us, Vs, H, W, patch_size, chroma)[0]


def rgbspace_dims(H, W):
    """(Hp, Wp, M) of the RGB colour-space branch: reflect padding to multiples of 8, one row per 8x8 patch."""
    Hp, Wp = H + (8 - H % 8) % 8, W + (8 - W % 8) % 8
    return Hp, Wp, (Hp // 8) * (Wp // 8)


def _qmf_encode_rgbspace(ctx, dev, rank, quality, bounds, int_bounds, patch_size, num_iters, init_sign, init):
    """qmf_encode(color_space="RGB") (lrf/compression/qmf.py:164-212, 288-290): with patches one [M, 3 p q] matrix per image,
    without (patch_size None) the three channel planes as matrices [3, H, W].  8x8 patches with num_iters >= 1 run on the
    fused entry point; the other forms go matrix -> lrf_qmf_decompose_f32 / _bcd_f32 / _svd_init_f32 (any shape)."""
    if isinstance(rank, (list, tuple)) or isinstance(quality, (list, tuple)):
        raise ValueError("color_space='RGB' takes a scalar rank / quality")
    H, W = dev.shape[-2:]
    Hp, Wp, M, N = _lib.rgbspace_dims_any(H, W, patch_size)
    if rank is None:
        assert quality >= 0 and quality <= 100, "'quality' must be between 0 and 100."
        R = max(round(min(M, N) * quality / 100), 1)
    else:
        R = rank
    nmat = 1 if patch_size is not None else 3
    sign = None if init_sign is None else torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1, R).expand(nmat, R).contiguous()
    metadata = {"dtype": str(dev.dtype).split(".")[-1], "color space": "RGB", "patch": patch_size is not None, "bounds": bounds}
    if patch_size is not None:
        metadata.update({"patch size": patch_size, "original size": [H, W], "padded size": [Hp, Wp], "rank": R})
    else:
        metadata["rank"] = R
    if patch_size is not None and tuple(patch_size) == (8, 8) and num_iters >= 1:
        U, V = ctx.qmf_rgbspace_encode(dev, R, num_iters, int_bounds, sign, init)
        factors = [U[0].cpu().numpy(), V[0].cpu().numpy()]
    else:
        X = ctx.rgbspace_matrix_any(dev, patch_size)
        X = X[0] if patch_size is None else X  # [3, H, W]: three matrices; [1, M, N]: one
        if init is not None:
            u0 = torch.as_tensor(init[0], dtype=torch.float32).reshape(nmat, M, R).contiguous().cuda(dev.device)
            v0 = torch.as_tensor(init[1], dtype=torch.float32).reshape(nmat, N, R).contiguous().cuda(dev.device)
        elif num_iters == 0 or init is None:
            u0 = v0 = None
        sg = None if sign is None else sign.cuda(dev.device)
        if num_iters == 0:  # the float factors go straight through torch's truncating cast (qmf.py:188, 210)
            if init is None:
                u0, v0 = ctx.svd_init(X, R, sg)
            u, v = u0.cpu().to(torch.int8), v0.cpu().to(torch.int8)
        elif init is not None:
            u, v = ctx.bcd(X, u0, v0, num_iters, int_bounds[0], int_bounds[1])
        else:
            u, v = ctx.decompose(X, R, num_iters, int_bounds[0], int_bounds[1], sg)
        u, v = u.cpu().numpy(), v.cpu().numpy()
        factors = [u[0], v[0]] if patch_size is not None else [u, v]  # patch=False keeps the channel axis (qmf.py:208-210)
    return combine_bytes([dict_to_bytes(metadata), combine_bytes([encode_tensor(np.ascontiguousarray(f)) for f in factors])])


def _svd_init_factors(ctx, dev, ranks, init_sign):
    """num_iters=0 (experiments/ablation_numiters/eval.py:51): the float SVD factors go straight through `.to(int8)`
    (lrf/compression/qmf.py:258-260), i.e. torch's truncating cast, done here with torch itself on the host."""
    H, W = dev.shape[-2:]
    X = ctx.planes_from_rgb(dev)[0]
    factors, off, soff = [], 0, 0
    for (_, _, _, _, M), R in zip(_lib.plane_dims(H, W), ranks):
        x = X[off:off + M * 64].reshape(1, M, 64)
        off += M * 64
        sign = None
        if init_sign is not None:
            sign = torch.as_tensor(init_sign, dtype=torch.int8).reshape(-1)[soff:soff + R].reshape(1, R).contiguous().cuda(dev.device)
        soff += R
        u0, v0 = ctx.svd_init(x, R, sign)
        factors += [u0[0].cpu().to(torch.int8).numpy(), v0[0].cpu().to(torch.int8).numpy()]
    return factors


def parse_stream(encoded_image: bytes):
    """-> (metadata, [u_y, v_y, u_cb, v_cb, u_cr, v_cr]) for the YCbCr branch (lrf/compression/qmf.py:306-327)."""
    encoded_metadata, encoded_factors = separate_bytes(encoded_image, 2)
    metadata = bytes_to_dict(encoded_metadata)
    if metadata["color space"] != "YCbCr" or not metadata["patch"] or list(metadata["patch size"]) != [8, 8]:
        raise NotImplementedError("HIP decode covers the YCbCr / 8x8-patch branch only")
    factors = [decode_tensor(f) for f in separate_bytes(encoded_factors, 6)]
    return metadata, factors


def _factors_python(streams: Sequence[bytes]):
    """The streams' factors by this module's own container code, every shape checked against the stream's metadata
    -> (metadata of each stream, U [B, sum M_c R_c], V [B, 64 sum R_c]) as int8 arrays."""
    metas, Us, Vs = [], [], []
    for s in streams:
        meta, f = parse_stream(s)
        # a truncated or crafted stream must not reach the kernels: they index the factors from the metadata alone
        H0, W0 = meta["original size"][0]
        ranks0 = [int(r) for r in meta["rank"]]
        if len(ranks0) != 3 or min(ranks0) < 1:
            raise ValueError("stream metadata: 'rank' must hold three positive integers")
        for c, (d, R) in enumerate(zip(_lib.plane_dims(H0, W0), ranks0)):
            u, v = f[2 * c], f[2 * c + 1]
            if tuple(u.shape) != (d[4], R) or tuple(v.shape) != (64, R) or u.dtype != np.int8 or v.dtype != np.int8:
                raise ValueError(f"stream factors of plane {c} are {u.dtype}{tuple(u.shape)} / {v.dtype}{tuple(v.shape)}; "
                                 f"the metadata describes int8 {(d[4], R)} / {(64, R)}")
        metas.append(meta)
        Us.append(np.concatenate([np.ascontiguousarray(f[i], dtype=np.int8).ravel() for i in (0, 2, 4)]))
        Vs.append(np.concatenate([np.ascontiguousarray(f[i], dtype=np.int8).ravel() for i in (1, 3, 5)]))
    for m in metas[1:]:
        if m["original size"] != metas[0]["original size"] or m["rank"] != metas[0]["rank"]:
            raise ValueError("streams differ in geometry or ranks")
    return metas, np.stack(Us), np.stack(Vs)


def _factors_native(streams: Sequence[bytes]):
    """The same through liblrf_pack.so (lrf_pack_unpack_qmf_factors: the columns of all streams inflate on host threads; one
    stream 0.21 -> ~0.05 ms, 256 streams 51 -> ~1 ms), or None when the library is absent or refuses a stream — anything
    that is not exactly the int8 / column layout — so that _factors_python can say what is wrong with it."""
    import ctypes
    try:
        lib = _pack_lib()
    except OSError:
        return None
    metas, blobs = [], []
    for s in streams:
        try:
            encoded_metadata, encoded_factors = separate_bytes(s, 2)
            meta = bytes_to_dict(encoded_metadata)
            H0, W0 = meta["original size"][0]
            ranks = [int(r) for r in meta["rank"]]
            ok = (meta["color space"] == "YCbCr" and meta["patch"] and list(meta["patch size"]) == [8, 8] and len(ranks) == 3
                  and min(ranks) >= 1 and int(H0) >= 1 and int(W0) >= 1)
        except Exception:  # malformed: the Python path raises the proper error
            return None
        if not ok or (metas and (meta["original size"] != metas[0]["original size"] or meta["rank"] != metas[0]["rank"])):
            return None
        metas.append(meta)
        blobs.append(encoded_factors)
    H, W = metas[0]["original size"][0]
    ranks = [int(r) for r in metas[0]["rank"]]
    if max(ranks) > 64:
        return None
    dims = _lib.plane_dims(int(H), int(W))
    B = len(streams)
    # the metadata is untrusted and nothing of the payload has been validated yet: a deflate stream expands by at most
    # ~1032x, so a stream that claims more factor bytes than its payload could inflate to is refused before the buffers
    # it asks for are allocated (the Python parser then names the defect, or raises the same way)
    need = sum(d[4] * r for d, r in zip(dims, ranks)) + 64 * sum(ranks)
    if any(need > 1040 * len(b) + 4096 for b in blobs):
        return None
    for m in metas:  # what reaches the kernels is the validated integer form, not the raw JSON values
        m["rank"] = list(ranks)
    U = np.empty((B, sum(d[4] * r for d, r in zip(dims, ranks))), dtype=np.int8)
    V = np.empty((B, 64 * sum(ranks)), dtype=np.int8)
    ptrs = (ctypes.c_char_p * B)(*blobs)
    lens = (ctypes.c_int64 * B)(*[len(b) for b in blobs])
    M = (ctypes.c_int64 * 3)(*[d[4] for d in dims])
    R = (ctypes.c_int * 3)(*ranks)
    rc = lib.lrf_pack_unpack_qmf_factors(ptrs, lens, B, M, R, 0, U.ctypes.data_as(ctypes.c_void_p), U.shape[1],
                                         V.ctypes.data_as(ctypes.c_void_p), V.shape[1])
    return (metas, U, V) if rc == 0 else None


def qmf_decode_batch(streams: Sequence[bytes], device=None) -> torch.Tensor:
    """Decodes streams of equal geometry and ranks -> uint8 CUDA tensor [B,3,H,W].  Streams of the other branches (another
    patch size, patch=False, another chroma scale, the RGB colour space) are decoded one by one and stacked."""
    m_first = bytes_to_dict(separate_bytes(streams[0], 2)[0])
    H0, W0 = (m_first["original size"][0] if m_first["color space"] == "YCbCr" else (0, 0))
    if m_first["color space"] != "YCbCr" or not m_first["patch"] or list(m_first["patch size"]) != [8, 8] \
            or list(m_first["original size"][1]) != [H0 // 2, W0 // 2]:
        one = _qmf_decode_rgbspace if m_first["color space"] == "RGB" else _qmf_decode_anyshape
        return torch.stack([one(s) for s in streams])
    got = _factors_native(streams)
    metas, Uh, Vh = got if got is not None else _factors_python(streams)
    m0 = metas[0]
    for m in metas[1:]:
        if m["original size"] != m0["original size"] or m["rank"] != m0["rank"]:
            raise ValueError("streams differ in geometry or ranks")
    H, W = m0["original size"][0]
    dims = _lib.plane_dims(H, W)
    for c in range(3):  # the stream must describe the geometry the kernels derive from (H, W)
        if list(m0["original size"][c]) != [dims[c][0], dims[c][1]] or list(m0["padded size"][c]) != [dims[c][2], dims[c][3]]:
            raise NotImplementedError("stream geometry is not the scale_factor=(0.5,0.5) / 8x8 layout")
    ctx = _lib.context(device)
    U = torch.from_numpy(Uh).cuda(ctx.device)
    V = torch.from_numpy(Vh).cuda(ctx.device)
    if m0["dtype"] != "uint8":
        raise NotImplementedError("HIP decode writes uint8 images")
    return ctx.decode_rgb(U, V, H, W, m0["rank"])


def _qmf_decode_rgbspace(encoded_image: bytes, device=None) -> torch.Tensor:
    """RGB colour-space branch of qmf_decode (lrf/compression/qmf.py:309-323) -> uint8 CUDA tensor [3,H,W]."""
    encoded_metadata, encoded_factors = separate_bytes(encoded_image, 2)
    metadata = bytes_to_dict(encoded_metadata)
    if metadata["dtype"] != "uint8":
        raise NotImplementedError("HIP decode writes uint8 images")
    u, v = (decode_tensor(f) for f in separate_bytes(encoded_factors, 2))
    if u.dtype != np.int8 or v.dtype != np.int8:
        raise ValueError("stream factors are not int8")
    R = int(metadata["rank"])
    if metadata["patch"]:
        patch_size = tuple(metadata["patch size"])
        H, W = metadata["original size"]
        Hp, Wp, M, N = _lib.rgbspace_dims_any(H, W, patch_size)
        if list(metadata["padded size"]) != [Hp, Wp] or tuple(u.shape) != (M, R) or tuple(v.shape) != (N, R):
            raise ValueError("stream factors do not match the reflect-padded patch geometry its metadata describes")
    else:  # patch=False: u [3, H, R], v [3, W, R]; the image size is the factors' (qmf.py:320-323)
        patch_size = None
        if u.ndim != 3 or v.ndim != 3 or u.shape[0] != 3 or v.shape[0] != 3 or u.shape[2] != R or v.shape[2] != R:
            raise ValueError("stream factors are not [3, H, R] / [3, W, R]")
        H, W = int(u.shape[1]), int(v.shape[1])
    ctx = _lib.context(device)
    U = torch.from_numpy(np.array(u, dtype=np.int8)).unsqueeze(0).cuda(ctx.device)  # copies: decode_tensor may return read-only views
    V = torch.from_numpy(np.array(v, dtype=np.int8)).unsqueeze(0).cuda(ctx.device)
    if patch_size == (8, 8):
        return ctx.qmf_rgbspace_decode(U, V, H, W)[0]
    return ctx.qmf_rgbspace_decode_any(U, V, H, W, patch_size)[0]


def qmf_decode(encoded_image: bytes) -> torch.Tensor:
    """Decode a QMF stream -> uint8 tensor [3,H,W] on the CPU, like the reference's lrf.qmf_decode
    (lrf/compression/qmf.py:295-353)."""
    meta = bytes_to_dict(separate_bytes(encoded_image, 2)[0])
    if meta["color space"] == "RGB":
        return _qmf_decode_rgbspace(encoded_image).cpu()
    H, W = meta["original size"][0]
    if not meta["patch"] or list(meta["patch size"]) != [8, 8] or list(meta["original size"][1]) != [H // 2, W // 2]:
        return _qmf_decode_anyshape(encoded_image).cpu()
    return qmf_decode_batch([encoded_image])[0].cpu()
