"""ctypes binding of liblrf_hip.so (include/lrf_hip.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is visible, the calls raise."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblrf_hip.so")

LRF_MAX_RANK = 64
LRF_K_PLANES, LRF_K_INIT, LRF_K_BCD, LRF_K_VUPDATE, LRF_K_DECODE, LRF_K_GRAM, LRF_K_BCD_PERSIST, LRF_K_PLANES_GRAM = range(8)
KERNEL_NAMES = {LRF_K_PLANES: "k_planes", LRF_K_GRAM: "k_gram", LRF_K_INIT: "k_init", LRF_K_BCD: "k_bcd",
                LRF_K_VUPDATE: "k_vupdate", LRF_K_DECODE: "k_decode", LRF_K_BCD_PERSIST: "k_bcd_persist",
                LRF_K_PLANES_GRAM: "k_planes_gram"}

_lib = None
_lock = threading.Lock()

c_void_p, c_int, c_i64, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t


class LrfError(RuntimeError):
    pass


class QmfOpts(ctypes.Structure):
    """lrf_qmf_opts (include/lrf_hip.h)"""
    _fields_ = [("bounded", c_int), ("lo", ctypes.c_float), ("hi", ctypes.c_float), ("l2_u", ctypes.c_double), ("l2_v", ctypes.c_double),
                ("l1_ratio", ctypes.c_double), ("factors", c_int), ("eps", ctypes.c_double), ("w_init", c_int)]


def load():
    """Loads liblrf_hip.so; raises ImportError when it has not been built (see __graft_entry__.build)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build the HIP extension first "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        lib.lrf_last_error.restype = ctypes.c_char_p
        lib.lrf_ctx_workspace_bytes.restype = c_size_t
        lib.lrf_ctx_workspace_bytes.argtypes = [c_void_p]
        lib.lrf_ctx_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
        lib.lrf_ctx_destroy.argtypes = [c_void_p]
        lib.lrf_ctx_destroy.restype = None
        lib.lrf_ctx_set_stream.argtypes = [c_void_p, c_void_p]
        lib.lrf_ctx_use_own_stream.argtypes = [c_void_p]
        lib.lrf_ctx_synchronize.argtypes = [c_void_p]
        lib.lrf_ctx_check.argtypes = [c_void_p]
        lib.lrf_ctx_trim.argtypes = [c_void_p]
        lib.lrf_ctx_profile.argtypes = [c_void_p, c_int]
        lib.lrf_ctx_profile_reset.argtypes = [c_void_p]
        lib.lrf_ctx_profile_kernels.argtypes = [c_void_p, ctypes.c_uint]
        lib.lrf_ctx_kernel_time.argtypes = [c_void_p, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        lib.lrf_malloc.argtypes = [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]
        lib.lrf_free.argtypes = [c_void_p, c_void_p]
        lib.lrf_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
        lib.lrf_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
        lib.lrf_plane_dims.argtypes = [c_i64, c_i64, c_int] + [ctypes.POINTER(c_i64)] * 5
        lib.lrf_qmf_planes_from_rgb_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p]
        lib.lrf_qmf_decompose_f32.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int,
                                              c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_bcd_f32.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_decompose_ex_f32.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, ctypes.POINTER(QmfOpts), c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_svd_init_f32.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_loss_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p]
        lib.lrf_qmf_encode_rgb_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, ctypes.POINTER(c_int), c_int, c_int,
                                              c_int, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_encode_sweep_rgb_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, ctypes.POINTER(c_int), c_int, c_int,
                                                    c_int, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_decode_rgb_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, ctypes.POINTER(c_int),
                                              c_void_p]
        lib.lrf_svd_encode_rgb_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.lrf_svd_decode_rgb_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p]
        lib.lrf_qmf_rgbspace_encode_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                                   c_void_p, c_void_p, c_void_p, c_void_p]
        lib.lrf_qmf_rgbspace_decode_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p]
        lib.lrf_quantize_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p]
        lib.lrf_svd_decode_any_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_void_p]
        lib.lrf_rgbspace_dims_any.argtypes = [c_i64, c_i64, c_int, c_int] + [ctypes.POINTER(c_i64)] * 4
        lib.lrf_qmf_rgbspace_matrix_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_void_p]
        lib.lrf_qmf_rgbspace_decode_any_u8.argtypes = [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_void_p]
        lib.lrf_plane_dims_any.argtypes = [c_i64, c_i64, c_int, c_int, c_int] + [ctypes.POINTER(c_i64)] * 6
        lib.lrf_plane_dims_any_hw.argtypes = [c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int] + [ctypes.POINTER(c_i64)] * 6
        lib.lrf_qmf_planes_any_hw_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_void_p]
        lib.lrf_qmf_decode_any_hw_u8.argtypes = [c_void_p] + [c_void_p] * 6 + [c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int,
                                                                              ctypes.POINTER(c_int), c_void_p]
        lib.lrf_qmf_planes_any_u8.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_void_p]
        lib.lrf_qmf_decode_any_u8.argtypes = [c_void_p] + [c_void_p] * 6 + [c_i64, c_i64, c_i64, c_int, c_int,
                                                                           ctypes.POINTER(c_int), c_void_p]
        lib.lrf_pipe_create.argtypes = [c_int, c_int, c_i64, ctypes.POINTER(c_void_p)]
        lib.lrf_pipe_destroy.argtypes = [c_void_p]
        lib.lrf_pipe_destroy.restype = None
        lib.lrf_pipe_slots.argtypes = [c_void_p]
        lib.lrf_pipe_slot_ctx.argtypes = [c_void_p, c_int]
        lib.lrf_pipe_slot_ctx.restype = c_void_p
        lib.lrf_pipe_workspace_bytes.argtypes = [c_void_p]
        lib.lrf_pipe_workspace_bytes.restype = c_size_t
        lib.lrf_pipe_qmf_encode_rgb_u8_host.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, ctypes.POINTER(c_int), c_int,
                                                        c_int, c_int, c_void_p, c_void_p, c_void_p]
        lib.lrf_pipe_qmf_encode_submit.argtypes = [c_void_p, c_void_p, c_i64, c_i64, c_i64, ctypes.POINTER(c_int), c_int, c_int,
                                                   c_int, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int)]
        lib.lrf_pipe_wait_next.argtypes = [c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
        lib.lrf_host_alloc.argtypes = [c_size_t, ctypes.POINTER(c_void_p)]
        lib.lrf_host_free.argtypes = [c_void_p]
        lib.lrf_host_register.argtypes = [c_void_p, c_size_t]
        lib.lrf_host_unregister.argtypes = [c_void_p]
        _lib = lib
        return lib


EXPORTS = ["lrf_last_error", "lrf_device_count", "lrf_version", "lrf_ctx_create", "lrf_ctx_destroy", "lrf_ctx_set_stream", "lrf_ctx_use_own_stream",
           "lrf_ctx_synchronize", "lrf_ctx_check", "lrf_ctx_workspace_bytes", "lrf_ctx_trim", "lrf_ctx_profile", "lrf_ctx_profile_kernels", "lrf_ctx_kernel_time",
           "lrf_ctx_profile_reset", "lrf_malloc", "lrf_free", "lrf_memcpy_h2d", "lrf_memcpy_d2h", "lrf_plane_dims",
           "lrf_qmf_planes_from_rgb_u8", "lrf_qmf_decompose_f32", "lrf_qmf_decompose_ex_f32", "lrf_qmf_bcd_f32", "lrf_qmf_svd_init_f32", "lrf_qmf_loss_f32",
           "lrf_qmf_encode_rgb_u8", "lrf_qmf_encode_sweep_rgb_u8", "lrf_qmf_decode_rgb_u8", "lrf_svd_encode_rgb_u8", "lrf_svd_decode_rgb_u8",
           "lrf_qmf_rgbspace_encode_u8", "lrf_qmf_rgbspace_decode_u8", "lrf_rgbspace_dims_any", "lrf_qmf_rgbspace_matrix_u8",
           "lrf_qmf_rgbspace_decode_any_u8", "lrf_quantize_u8", "lrf_svd_decode_any_u8",
           "lrf_plane_dims_any", "lrf_qmf_planes_any_u8", "lrf_qmf_decode_any_u8", "lrf_plane_dims_any_hw", "lrf_qmf_planes_any_hw_u8",
           "lrf_qmf_decode_any_hw_u8",
           "lrf_pipe_create", "lrf_pipe_destroy", "lrf_pipe_slots", "lrf_pipe_slot_ctx", "lrf_pipe_workspace_bytes",
           "lrf_pipe_qmf_encode_rgb_u8_host", "lrf_pipe_qmf_encode_submit", "lrf_pipe_wait_next",
           "lrf_host_alloc", "lrf_host_free", "lrf_host_register", "lrf_host_unregister"]


def check(rc):
    if rc == 0:
        return
    msg = load().lrf_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -2:
        raise NotImplementedError(msg)
    if rc == -4:
        raise MemoryError(msg)
    raise LrfError(msg)


def plane_dims(H, W):
    """[(h, w, hp, wp, M)] of the Y, Cb, Cr planes of an H x W image (host-only arithmetic)."""
    out = []
    for c in range(3):
        v = [c_i64() for _ in range(5)]
        check(load().lrf_plane_dims(H, W, c, *[ctypes.byref(x) for x in v]))
        out.append(tuple(int(x.value) for x in v))
    return out


def plane_dims_any(H, W, patch_size, chroma=None):
    """[(h, w, hp, wp, M, N)] of the Y, Cb, Cr planes for patches (p, q); patch_size None = patch=False (M, N = h, w).
    chroma: (hc, wc) for a scale_factor other than (0.5, 0.5); None = floor(H / 2) x floor(W / 2)."""
    p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
    hc, wc = (0, 0) if chroma is None else (int(chroma[0]), int(chroma[1]))
    out = []
    for c in range(3):
        v = [c_i64() for _ in range(6)]
        check(load().lrf_plane_dims_any_hw(H, W, hc, wc, p, q, c, *[ctypes.byref(x) for x in v]))
        out.append(tuple(int(x.value) for x in v))
    return out


def chroma_size(H, W, scale_factor):
    """floor(H * s_h) x floor(W * s_w): what F.interpolate(scale_factor=...) produces (lrf/compression/utils.py:92-94); None for
    the default (0.5, 0.5)"""
    import math
    if tuple(scale_factor) == (0.5, 0.5):
        return None
    return (int(math.floor(float(H) * float(scale_factor[0]))), int(math.floor(float(W) * float(scale_factor[1]))))


def rgbspace_dims_any(H, W, patch_size):
    """(Hp, Wp, M, N) of the RGB colour-space branch for patches (p, q); patch_size None = patch=False (per channel [H, W])."""
    p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
    v = [c_i64() for _ in range(4)]
    check(load().lrf_rgbspace_dims_any(H, W, p, q, *[ctypes.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)


def _dptr(t):
    """device pointer of a torch CUDA tensor (must be contiguous) or None"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous CUDA tensor"
    return c_void_p(t.data_ptr())


class Context:
    """One lrf_ctx: a device, a stream and the scratch workspace.  Not thread-safe."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = c_void_p()
        check(self._lib.lrf_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            self._lib.lrf_ctx_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self):
        import torch
        check(self._lib.lrf_ctx_set_stream(self._h, c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def synchronize(self):
        check(self._lib.lrf_ctx_synchronize(self._h))

    def check(self):
        """Raises LrfError if a persistent launch (k_bcd_p) on this context gave up since the last look (include/lrf_hip.h,
        lrf_ctx_check).  Does not wait: call it once the stream has been waited for — after `.cpu()` of a result, a
        torch.cuda.synchronize().  Every wrapper that hands results to the host calls it (`to_host`)."""
        check(self._lib.lrf_ctx_check(self._h))

    def to_host(self, *tensors):
        """the tensors on the host (torch's copy waits for the stream the kernels ran on), then `check`: the one way results
        of the encoder leave the device in this package, so that a failed launch raises in the call it belongs to"""
        out = tuple(t.cpu() for t in tensors)
        self.check()
        return out

    def workspace_bytes(self):
        return int(self._lib.lrf_ctx_workspace_bytes(self._h))

    def trim(self):
        """waits for the stream and releases the scratch workspace (it is re-grown by the next call)"""
        check(self._lib.lrf_ctx_trim(self._h))

    def profile(self, enable=True):
        check(self._lib.lrf_ctx_profile(self._h, int(bool(enable))))

    def profile_kernels(self, kernel_ids):
        """event-time only these kernel ids (an iterable of LRF_K_* numbers; empty = off)"""
        mask = 0
        for k in kernel_ids:
            mask |= 1 << int(k)
        check(self._lib.lrf_ctx_profile_kernels(self._h, mask))

    def profile_reset(self):
        check(self._lib.lrf_ctx_profile_reset(self._h))

    def kernel_time(self, kernel_id):
        ms, n = ctypes.c_double(), ctypes.c_long()
        check(self._lib.lrf_ctx_kernel_time(self._h, kernel_id, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    # ---- hot path (torch CUDA tensors in / out) ----
    def planes_from_rgb(self, rgb):
        import torch
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        floats = sum(d[4] for d in plane_dims(H, W)) * 64
        X = torch.empty((B, floats), dtype=torch.float32, device=rgb.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_planes_from_rgb_u8(self._h, _dptr(rgb), B, H, W, _dptr(X)))
        return X

    def decompose(self, X, R, K, lo, hi, sign=None):
        import torch
        X = X.contiguous()
        B, M, N = X.shape
        U = torch.empty((B, M, R), dtype=torch.int8, device=X.device)
        V = torch.empty((B, N, R), dtype=torch.int8, device=X.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_decompose_f32(self._h, _dptr(X), B, M, N, R, K, lo, hi, _dptr(sign), _dptr(U), _dptr(V)))
        return U, V

    def decompose_ex(self, X, R, K, bounds=(None, None), l2=0.0, l1_ratio=0.0, factor=(0, 1, 2), sign=None, init=None, eps=1e-16, w_init=None):
        """The general QMF.decompose (lrf_qmf_decompose_ex_f32): X [B,M,N] fp32 CUDA -> fp32 (U [B,M,R], V [B,N,R], W [B,2]).
        init = (u0, v0): initial factors instead of the library's SVD; w_init [B,2]: the initial affine pair that belongs to
        them (SVDInit(num_levels=...)); eps: CoordinateDescent's eps."""
        import torch
        X = X.float().contiguous()
        B, M, N = X.shape
        U = torch.empty((B, M, R), dtype=torch.float32, device=X.device)
        V = torch.empty((B, N, R), dtype=torch.float32, device=X.device)
        W = torch.empty((B, 2), dtype=torch.float32, device=X.device)
        bounded = bounds is not None and tuple(bounds) != (None, None)
        l2 = tuple(l2) if isinstance(l2, (tuple, list)) else (l2, l2)
        opts = QmfOpts(int(bounded), float(bounds[0]) if bounded else 0.0, float(bounds[1]) if bounded else 0.0, float(l2[0]), float(l2[1]),
                       float(l1_ratio), sum(1 << int(f) for f in set(factor)), 0.0 if eps == 1e-16 else float(eps), int(w_init is not None))
        if w_init is not None:
            assert init is not None, "w_init belongs to initial factors"
            W.copy_(w_init.to(device=X.device, dtype=torch.float32).reshape(B, 2))
        u0 = v0 = None
        if init is not None:
            u0, v0 = (t.to(device=X.device, dtype=torch.float32).contiguous() for t in init)
            assert tuple(u0.shape) == (B, M, R) and tuple(v0.shape) == (B, N, R)
        if sign is not None:
            sign = sign.to(device=X.device, dtype=torch.int8).contiguous()
        self.use_torch_stream()
        check(self._lib.lrf_qmf_decompose_ex_f32(self._h, _dptr(X), B, M, N, int(R), int(K), ctypes.byref(opts), _dptr(sign), _dptr(u0),
                                                 _dptr(v0), _dptr(U), _dptr(V), _dptr(W)))
        return U, V, W

    def bcd(self, X, U0, V0, K, lo, hi):
        import torch
        X, U0, V0 = X.contiguous(), U0.float().contiguous(), V0.float().contiguous()
        B, M, N = X.shape
        R = U0.shape[-1]
        U = torch.empty((B, M, R), dtype=torch.int8, device=X.device)
        V = torch.empty((B, N, R), dtype=torch.int8, device=X.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_bcd_f32(self._h, _dptr(X), B, M, N, R, K, lo, hi, _dptr(U0), _dptr(V0), _dptr(U), _dptr(V)))
        return U, V

    def svd_init(self, X, R, sign=None):
        import torch
        X = X.contiguous()
        B, M, N = X.shape
        U0 = torch.empty((B, M, R), dtype=torch.float32, device=X.device)
        V0 = torch.empty((B, N, R), dtype=torch.float32, device=X.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_svd_init_f32(self._h, _dptr(X), B, M, N, R, _dptr(sign), _dptr(U0), _dptr(V0)))
        return U0, V0

    def loss(self, X, U, V, W=None):
        """QMF.loss per matrix (lrf_qmf_loss_f32): X [B,M,N], U [B,M,R], V [B,N,R] fp32 CUDA, W [B,2] or None -> fp32 [B]"""
        import torch
        X, U, V = X.float().contiguous(), U.float().contiguous(), V.float().contiguous()
        B, M, N = X.shape
        R = U.shape[-1]
        assert tuple(U.shape) == (B, M, R) and tuple(V.shape) == (B, N, R)
        if W is not None:
            W = W.to(device=X.device, dtype=torch.float32).reshape(B, 2).contiguous()
        out = torch.empty((B,), dtype=torch.float32, device=X.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_loss_f32(self._h, _dptr(X), _dptr(U), _dptr(V), _dptr(W), B, M, N, int(R), _dptr(out)))
        return out

    def encode_rgb(self, rgb, ranks, K, lo, hi, sign=None, out=None):
        """rgb uint8 [B,3,H,W] (CUDA) -> (U int8 [B, sum M_c R_c], V int8 [B, 64 sum R_c])"""
        import torch
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        dims = plane_dims(H, W)
        nu = sum(d[4] * r for d, r in zip(dims, ranks))
        nv = 64 * sum(ranks)
        if out is None:
            U = torch.empty((B, nu), dtype=torch.int8, device=rgb.device)
            V = torch.empty((B, nv), dtype=torch.int8, device=rgb.device)
        else:  # the kernels write B * nu / B * nv bytes: anything else would be an out-of-bounds device write
            U, V = out
            for t, n, name in ((U, nu, "U"), (V, nv, "V")):
                if not (t.is_cuda and t.device == rgb.device and t.dtype == torch.int8 and t.is_contiguous()
                        and tuple(t.shape) == (B, n)):
                    raise ValueError(f"out {name} must be a contiguous int8 tensor of shape {(B, n)} on {rgb.device}")
        R = (c_int * 3)(*[int(r) for r in ranks])
        self.use_torch_stream()
        check(self._lib.lrf_qmf_encode_rgb_u8(self._h, _dptr(rgb), B, H, W, R, K, lo, hi, _dptr(sign), _dptr(U), _dptr(V)))
        return U, V

    def encode_sweep_rgb(self, rgb, triples, K, lo, hi, sign=None):
        """rgb uint8 [B,3,H,W] (CUDA) at every rank triple of `triples` in ONE call (lrf_qmf_encode_sweep_rgb_u8) ->
        [(U int8 [B, sum M_c R_c], V int8 [B, 64 sum R_c]) per triple] (views of two flat buffers)"""
        import torch
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8 and len(triples) >= 1
        dims = plane_dims(H, W)
        nus = [sum(d[4] * int(r) for d, r in zip(dims, t)) for t in triples]
        nvs = [64 * sum(int(r) for r in t) for t in triples]
        U = torch.empty((B * sum(nus),), dtype=torch.int8, device=rgb.device)
        V = torch.empty((B * sum(nvs),), dtype=torch.int8, device=rgb.device)
        R = (c_int * (3 * len(triples)))(*[int(r) for t in triples for r in t])
        self.use_torch_stream()
        check(self._lib.lrf_qmf_encode_sweep_rgb_u8(self._h, _dptr(rgb), B, H, W, len(triples), R, K, lo, hi, _dptr(sign), _dptr(U), _dptr(V)))
        out, uo, vo = [], 0, 0
        for nu, nv in zip(nus, nvs):
            out.append((U[uo:uo + B * nu].view(B, nu), V[vo:vo + B * nv].view(B, nv)))
            uo += B * nu
            vo += B * nv
        return out

    def decode_rgb(self, U, V, H, W, ranks):
        import torch
        U, V = U.contiguous(), V.contiguous()
        B = U.shape[0]
        # the kernel indexes the factors from (H, W) and the ranks alone: sizes that disagree would be out-of-bounds reads
        dims = plane_dims(H, W)
        nu, nv = sum(d[4] * int(r) for d, r in zip(dims, ranks)), 64 * sum(int(r) for r in ranks)
        if len(ranks) != 3 or U.dtype != torch.int8 or V.dtype != torch.int8 or tuple(U.shape) != (B, nu) or \
                tuple(V.shape) != (B, nv) or V.device != U.device:
            raise ValueError(f"factor buffers do not match the geometry: expected int8 U {(B, nu)} and V {(B, nv)}, "
                             f"got {U.dtype} {tuple(U.shape)} and {V.dtype} {tuple(V.shape)}")
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=U.device)
        R = (c_int * 3)(*[int(r) for r in ranks])
        self.use_torch_stream()
        check(self._lib.lrf_qmf_decode_rgb_u8(self._h, _dptr(U), _dptr(V), B, H, W, R, _dptr(rgb)))
        return rgb


    def planes_any(self, rgb, patch_size, ch, chroma=None):
        """rgb uint8 [B,3,H,W] (CUDA) -> X fp32 [B, M, N] of plane ch for patches (p, q) (None: the plane itself)"""
        import torch
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
        hc, wc = (0, 0) if chroma is None else (int(chroma[0]), int(chroma[1]))
        d = plane_dims_any(H, W, patch_size, chroma)[ch]
        X = torch.empty((B, d[4], d[5]), dtype=torch.float32, device=rgb.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_planes_any_hw_u8(self._h, _dptr(rgb.contiguous()), B, H, W, hc, wc, p, q, ch, _dptr(X)))
        return X

    def decode_any(self, Us, Vs, H, W, patch_size, chroma=None):
        """three (U [B,M_c,R_c], V [B,N_c,R_c]) int8 CUDA pairs -> uint8 [B,3,H,W]"""
        import torch
        Us = [u.contiguous() for u in Us]
        Vs = [v.contiguous() for v in Vs]
        B = Us[0].shape[0]
        p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
        hc, wc = (0, 0) if chroma is None else (int(chroma[0]), int(chroma[1]))
        dims = plane_dims_any(H, W, patch_size, chroma)
        for c in range(3):
            assert Us[c].shape[1] == dims[c][4] and Vs[c].shape[1] == dims[c][5] and Us[c].shape[2] == Vs[c].shape[2], \
                "factor shapes do not match the image geometry"
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=Us[0].device)
        R = (c_int * 3)(*[int(u.shape[2]) for u in Us])
        self.use_torch_stream()
        check(self._lib.lrf_qmf_decode_any_hw_u8(self._h, _dptr(Us[0]), _dptr(Vs[0]), _dptr(Us[1]), _dptr(Vs[1]), _dptr(Us[2]),
                                                 _dptr(Vs[2]), B, H, W, hc, wc, p, q, R, _dptr(rgb)))
        return rgb


def _svd_methods():
    def svd_encode_rgb(self, rgb, R, sign=None):
        """rgb uint8 [B,3,H,W] (CUDA) -> (U uint8 [B,M,R], V uint8 [B,192,R], qparams float [B,4])"""
        import torch
        rgb = rgb.contiguous()
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        M = ((H + 7) // 8) * ((W + 7) // 8)
        U = torch.empty((B, M, R), dtype=torch.uint8, device=rgb.device)
        V = torch.empty((B, 192, R), dtype=torch.uint8, device=rgb.device)
        qp = torch.empty((B, 4), dtype=torch.float32, device=rgb.device)
        self.use_torch_stream()
        check(self._lib.lrf_svd_encode_rgb_u8(self._h, _dptr(rgb), B, H, W, int(R), _dptr(sign), _dptr(U), _dptr(V), _dptr(qp)))
        return U, V, qp

    def svd_decode_rgb(self, U, V, qparams6, H, W):
        import torch
        U, V, qparams6 = U.contiguous(), V.contiguous(), qparams6.float().contiguous()
        B, _, R = U.shape
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=U.device)
        self.use_torch_stream()
        check(self._lib.lrf_svd_decode_rgb_u8(self._h, _dptr(U), _dptr(V), B, H, W, int(R), _dptr(qparams6), _dptr(rgb)))
        return rgb

    def qmf_rgbspace_encode(self, rgb, R, num_iters=10, bounds=(-16, 15), sign=None, init=None):
        """qmf_encode's RGB colour-space branch: rgb uint8 [B,3,H,W] (CUDA) -> (U int8 [B,M,R], V int8 [B,192,R]).
        init = (U0 [B,M,R], V0 [B,192,R]) fp32 overrides the SVD initialisation."""
        import torch
        rgb = rgb.contiguous()
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        M = ((H + 7) // 8) * ((W + 7) // 8)
        U = torch.empty((B, M, R), dtype=torch.int8, device=rgb.device)
        V = torch.empty((B, 192, R), dtype=torch.int8, device=rgb.device)
        u0 = v0 = None
        if init is not None:
            u0, v0 = (t.to(device=rgb.device, dtype=torch.float32).contiguous() for t in init)
            assert tuple(u0.shape) == (B, M, R) and tuple(v0.shape) == (B, 192, R)
        if sign is not None:
            sign = sign.to(device=rgb.device, dtype=torch.int8).contiguous()
        self.use_torch_stream()
        check(self._lib.lrf_qmf_rgbspace_encode_u8(self._h, _dptr(rgb), B, H, W, int(R), int(num_iters), int(bounds[0]), int(bounds[1]),
                                                  _dptr(sign), _dptr(u0), _dptr(v0), _dptr(U), _dptr(V)))
        return U, V

    def qmf_rgbspace_decode(self, U, V, H, W):
        import torch
        U, V = U.contiguous(), V.contiguous()
        B, _, R = U.shape
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=U.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_rgbspace_decode_u8(self._h, _dptr(U), _dptr(V), B, H, W, int(R), _dptr(rgb)))
        return rgb

    def rgbspace_matrix_any(self, rgb, patch_size):
        """rgb uint8 [B,3,H,W] (CUDA) -> X fp32: [B, M, 3 p q] for patches (p, q), [B, 3, H, W] for patch_size None"""
        import torch
        rgb = rgb.contiguous()
        B, C, H, W = rgb.shape
        assert C == 3 and rgb.dtype == torch.uint8
        p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
        _, _, M, N = rgbspace_dims_any(H, W, patch_size)
        X = torch.empty((B, 3, H, W) if patch_size is None else (B, M, N), dtype=torch.float32, device=rgb.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_rgbspace_matrix_u8(self._h, _dptr(rgb), B, H, W, p, q, _dptr(X)))
        return X

    def qmf_rgbspace_decode_any(self, U, V, H, W, patch_size):
        """int8 U [B,M,R] / V [B,3pq,R] (patches) or U [B,3,H,R] / V [B,3,W,R] (patch_size None) -> uint8 [B,3,H,W]"""
        import torch
        U, V = U.contiguous(), V.contiguous()
        B, R = U.shape[0], U.shape[-1]
        p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
        _, _, M, N = rgbspace_dims_any(H, W, patch_size)
        want_u, want_v = ((B, 3, H, R), (B, 3, W, R)) if patch_size is None else ((B, M, R), (B, N, R))
        if tuple(U.shape) != want_u or tuple(V.shape) != want_v or U.dtype != torch.int8 or V.dtype != torch.int8:
            raise ValueError(f"factor shapes {tuple(U.shape)} / {tuple(V.shape)} do not match the geometry {want_u} / {want_v}")
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=U.device)
        self.use_torch_stream()
        check(self._lib.lrf_qmf_rgbspace_decode_any_u8(self._h, _dptr(U), _dptr(V), B, H, W, p, q, int(R), _dptr(rgb)))
        return rgb

    def quantize_u8(self, T):
        """quantize(t, uint8) of each T[b] as a whole (utils.py:185-220): fp32 CUDA [B, ...] -> (uint8 same shape, qparams [B,2])"""
        import torch
        T = T.float().contiguous()
        B = T.shape[0]
        per = T[0].numel()
        Q = torch.empty(T.shape, dtype=torch.uint8, device=T.device)
        qp = torch.empty((B, 2), dtype=torch.float32, device=T.device)
        self.use_torch_stream()
        check(self._lib.lrf_quantize_u8(self._h, _dptr(T), B, per, _dptr(Q), _dptr(qp)))
        return Q, qp

    def svd_decode_any(self, U, V, H, W, patch_size, qparams6=None):
        """uint8 factors + qparams6 [B,6], or float32 factors (qparams6 None) -> uint8 [B,3,H,W]; layouts as qmf_rgbspace_decode_any"""
        import torch
        U, V = U.contiguous(), V.contiguous()
        B, R = U.shape[0], U.shape[-1]
        is_float = U.dtype == torch.float32
        assert U.dtype == V.dtype and (is_float or U.dtype == torch.uint8)
        p, q = (0, 0) if patch_size is None else (int(patch_size[0]), int(patch_size[1]))
        _, _, M, N = rgbspace_dims_any(H, W, patch_size)
        want_u, want_v = ((B, 3, H, R), (B, 3, W, R)) if patch_size is None else ((B, M, R), (B, N, R))
        if tuple(U.shape) != want_u or tuple(V.shape) != want_v:
            raise ValueError(f"factor shapes {tuple(U.shape)} / {tuple(V.shape)} do not match the geometry {want_u} / {want_v}")
        if not is_float:
            qparams6 = qparams6.float().contiguous()
            assert tuple(qparams6.shape) == (B, 6)
        rgb = torch.empty((B, 3, H, W), dtype=torch.uint8, device=U.device)
        self.use_torch_stream()
        check(self._lib.lrf_svd_decode_any_u8(self._h, _dptr(U), _dptr(V), int(is_float), B, H, W, p, q, int(R),
                                              None if is_float else _dptr(qparams6), _dptr(rgb)))
        return rgb

    Context.quantize_u8 = quantize_u8
    Context.svd_decode_any = svd_decode_any
    Context.rgbspace_matrix_any = rgbspace_matrix_any
    Context.qmf_rgbspace_decode_any = qmf_rgbspace_decode_any
    Context.svd_encode_rgb = svd_encode_rgb
    Context.svd_decode_rgb = svd_decode_rgb
    Context.qmf_rgbspace_encode = qmf_rgbspace_encode
    Context.qmf_rgbspace_decode = qmf_rgbspace_decode


_svd_methods()


class _BorrowedContext(Context):
    """A Context view of an lrf_ctx owned by something else (a pipe's slot): profiling calls only, never destroyed here."""

    def __init__(self, handle, device):
        self._lib = load()
        self._h = c_void_p(handle)
        self.device = int(device)

    def close(self):
        self._h = c_void_p()


def _hptr(t):
    """host pointer of a contiguous torch CPU tensor / numpy array, or None"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"], "expected a C-contiguous array"
        return c_void_p(t.ctypes.data)
    assert (not t.is_cuda) and t.is_contiguous(), "expected a contiguous CPU tensor"
    return c_void_p(t.data_ptr())


class Pipe:
    """lrf_pipe: the host -> host pipelined encoder (include/lrf_hip.h).  Host tensors in, int8 factors back on the
    host; sub-batches stream through `slots` independent encoder contexts so that uploads, kernels and downloads
    overlap.  Not thread-safe; one per (host thread, device)."""

    def __init__(self, device=0, slots=2, sub_batch=0):
        import torch
        if not torch.cuda.is_available():
            raise LrfError("lrf_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self._lib = load()
        self._h = c_void_p()
        self.device = int(device)
        check(self._lib.lrf_pipe_create(self.device, int(slots), int(sub_batch), ctypes.byref(self._h)))
        self.slots = int(self._lib.lrf_pipe_slots(self._h))

    def close(self):
        if self._h:
            self._lib.lrf_pipe_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot_context(self, slot):
        return _BorrowedContext(self._lib.lrf_pipe_slot_ctx(self._h, int(slot)), self.device)

    def workspace_bytes(self):
        return int(self._lib.lrf_pipe_workspace_bytes(self._h))

    def _prepare(self, rgb, ranks, sign, out):
        import torch
        assert (not rgb.is_cuda) and rgb.dtype == torch.uint8 and rgb.dim() == 4 and rgb.shape[1] == 3, \
            "expected a uint8 CPU tensor [B,3,H,W]"
        rgb = rgb.contiguous()
        B, _, H, W = rgb.shape
        dims = plane_dims(H, W)
        nu, nv = sum(d[4] * int(r) for d, r in zip(dims, ranks)), 64 * sum(int(r) for r in ranks)
        if out is None:
            pin = rgb.is_pinned()
            U = torch.empty((B, nu), dtype=torch.int8, pin_memory=pin)
            V = torch.empty((B, nv), dtype=torch.int8, pin_memory=pin)
        else:
            U, V = out
            for t, n, name in ((U, nu, "U"), (V, nv, "V")):
                if t.is_cuda or t.dtype != torch.int8 or not t.is_contiguous() or tuple(t.shape) != (B, n):
                    raise ValueError(f"out {name} must be a contiguous int8 CPU tensor of shape {(B, n)}")
        if sign is not None:
            sign = torch.as_tensor(sign, dtype=torch.int8).reshape(-1, sum(int(r) for r in ranks))
            sign = sign.expand(B, -1).contiguous()
        R = (c_int * 3)(*[int(r) for r in ranks])
        return rgb, B, H, W, R, sign, U, V

    def encode_rgb_host(self, rgb, ranks, K, lo, hi, sign=None, out=None):
        """rgb uint8 CPU tensor [B,3,H,W] (pinned for full speed) -> (U int8 [B, sum M_c R_c], V int8 [B, 64 sum R_c])
        CPU tensors; returns when they are complete."""
        rgb, B, H, W, R, sign, U, V = self._prepare(rgb, ranks, sign, out)
        check(self._lib.lrf_pipe_qmf_encode_rgb_u8_host(self._h, _hptr(rgb), B, H, W, R, int(K), int(lo), int(hi), _hptr(sign),
                                                        _hptr(U), _hptr(V)))
        return U, V

    def encode_rgb_host_iter(self, rgb, ranks, K, lo, hi, sign=None, out=None):
        """Generator form: enqueues the whole batch, then yields (first_image, n_images, U, V) as each sub-batch lands on
        the host (U, V are the full output tensors; rows [first, first + n) are final at that point)."""
        rgb, B, H, W, R, sign, U, V = self._prepare(rgb, ranks, sign, out)
        n_sub = c_int()
        rc = self._lib.lrf_pipe_qmf_encode_submit(self._h, _hptr(rgb), B, H, W, R, int(K), int(lo), int(hi), _hptr(sign), _hptr(U),
                                                  _hptr(V), ctypes.byref(n_sub))
        try:
            check(rc)
            while True:
                first, n = c_i64(), c_i64()
                check(self._lib.lrf_pipe_wait_next(self._h, ctypes.byref(first), ctypes.byref(n)))
                if n.value == 0:
                    return
                yield int(first.value), int(n.value), U, V
        finally:  # never leave copies into the caller's buffers in flight (an abandoned generator, an error)
            n = c_i64(1)
            for _ in range(max(1, int(n_sub.value)) + 1):  # (a piece that reports a failed launch still counts as waited for)
                self._lib.lrf_pipe_wait_next(self._h, None, ctypes.byref(n))
                if not n.value:
                    break


_contexts = {}
_pipes = {}


def pipe(device=None, slots=2, sub_batch=0) -> Pipe:
    """The cached pipe of the calling THREAD for (device, slots, sub_batch).  A Pipe is a per-(host thread, device) object
    (its slot buffers, submission cursor and sign table are not locked, and ctypes releases the GIL inside its calls), so
    two Python threads encoding at once each get a pipe of their own."""
    import threading

    import torch
    if not torch.cuda.is_available():
        raise LrfError("lrf_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device("cuda", device).index if not isinstance(device, int) else device
    key = (threading.get_ident(), device, int(slots), int(sub_batch))
    with _lock:
        p = _pipes.get(key)
    if p is None:
        p = Pipe(device, slots, sub_batch)
        with _lock:
            _pipes[key] = p
    return p


def context(device=None) -> Context:
    """The cached per-device context of the calling process."""
    import torch
    if not torch.cuda.is_available():
        raise LrfError("lrf_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device("cuda", device).index if not isinstance(device, int) else device
    with _lock:
        ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _contexts[device] = ctx
    return ctx
