"""ctypes front-end of the CPU oracle (oracle/lrf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
lrf_amd/ never does.  Arrays are numpy; nothing here touches torch or the GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LRF_ORACLE_SO: another build of the same sources (the sanitizer build of `make -C oracle san`, tools/run_sanitizers.sh)
_SO = os.environ.get("LRF_ORACLE_SO") or os.path.join(_HERE, "_build", "liblrf_oracle.so")
_lib = None

c_long = ctypes.c_long
c_int = ctypes.c_int
c_float = ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
_dp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i8p = ctypes.POINTER(ctypes.c_int8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lrf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def plane_dims(H, W, p=8, q=8):
    """[(h, w, hp, wp, M)] for Y, Cb, Cr."""
    out = []
    for c in range(3):
        v = [c_long() for _ in range(5)]
        lib().lrf_oracle_plane_dims(c_long(H), c_long(W), c_long(p), c_long(q), c_int(c), *[ctypes.byref(x) for x in v])
        out.append(tuple(int(x.value) for x in v))
    return out


def rgb_to_planes(rgb, p=8, q=8):
    """uint8 [3,H,W] -> [X_Y, X_Cb, X_Cr] fp32 [M_c, p*q]."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    dims = plane_dims(H, W, p, q)
    X = [np.empty((d[4], p * q), np.float32) for d in dims]
    rc = lib().lrf_oracle_rgb_to_planes(_ptr(rgb, _u8p), c_long(H), c_long(W), c_long(p), c_long(q),
                                        _ptr(X[0], _fp), _ptr(X[1], _fp), _ptr(X[2], _fp))
    assert rc == 0
    return X


def gram_f64(X):
    X = _f32(X)
    M, N = X.shape
    G = np.empty((N, N), np.float64)
    lib().lrf_oracle_gram_f64(_ptr(X, _fp), c_long(M), c_long(N), _ptr(G, _dp))
    return G


def gram_exact(X, E=None):
    """The exact Gram matrix of the 64-column path (rounded once to fp64) and the grid exponent used."""
    X = _f32(X)
    M, N = X.shape
    if E is None:
        E = int(lib().lrf_oracle_gram_exponent(_ptr(X, _fp), c_long(M * N)))
    G = np.empty((N, N), np.float64)
    lib().lrf_oracle_gram_exact(_ptr(X, _fp), c_long(M), c_long(N), c_int(E), _ptr(G, _dp))
    return G, E


def jacobi_f64(A, max_sweeps=30):
    A = np.array(A, dtype=np.float64, order="C")
    n = A.shape[0]
    E = np.empty((n, n), np.float64)
    lib().lrf_oracle_jacobi_f64.restype = c_int
    sweeps = lib().lrf_oracle_jacobi_f64(_ptr(A, _dp), c_int(n), _ptr(E, _dp), c_int(max_sweeps))
    return np.diag(A).copy(), E, sweeps


def top_eig_f64(G, R):
    """(lam[R] descending, E[64,R]) of a symmetric 64x64 matrix, by the oracle's tridiagonal path."""
    A = np.array(G, dtype=np.float64, order="C")
    lam = np.empty(R, np.float64)
    Ev = np.empty((R, 64), np.float64)
    lib().lrf_oracle_top_eig_f64.restype = c_int
    rc = lib().lrf_oracle_top_eig_f64(_ptr(A, _dp), c_int(R), _ptr(lam, _dp), _ptr(Ev, _dp))
    assert rc == 0
    return lam, Ev.T.copy()


def _sign_arg(sign, R):
    if sign is None:
        return None, None
    s = np.ascontiguousarray(sign, dtype=np.int8)
    assert s.shape == (R,)
    return s, _ptr(s, _i8p)


def svd_init(X, R, sign=None):
    X = _f32(X)
    M, N = X.shape
    u0 = np.empty((M, R), np.float32)
    v0 = np.empty((N, R), np.float32)
    keep, sp = _sign_arg(sign, R)
    rc = lib().lrf_oracle_svd_init(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), sp, _ptr(u0, _fp), _ptr(v0, _fp))
    assert rc == 0
    return u0, v0


def bcd(X, u0, v0, num_iters, bounds=(-16, 15)):
    """Runs num_iters BCD iterations from (u0, v0); returns fp32 (U, V)."""
    X = _f32(X)
    M, N = X.shape
    U = np.array(u0, dtype=np.float32, order="C")
    V = np.array(v0, dtype=np.float32, order="C")
    R = U.shape[1]
    assert U.shape == (M, R) and V.shape == (N, R)
    bounded = bounds is not None and tuple(bounds) != (None, None)
    lo, hi = (bounds if bounded else (0.0, 0.0))
    rc = lib().lrf_oracle_bcd(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), c_int(num_iters), c_int(int(bounded)),
                              c_float(lo), c_float(hi), _ptr(U, _fp), _ptr(V, _fp))
    assert rc == 0
    return U, V


def bcd_ex(X, u0, v0, num_iters, bounds=(None, None), l2=(0.0, 0.0), l1_ratio=0.0, factor=(0, 1, 2), w=(0.0, 1.0), eps=1e-16):
    """The general CoordinateDescent loop (lrf_oracle_bcd_ex): returns fp32 (U, V, W[2])."""
    X = _f32(X)
    M, N = X.shape
    U = np.array(u0, dtype=np.float32, order="C")
    V = np.array(v0, dtype=np.float32, order="C")
    R = U.shape[1]
    W = np.array(w, dtype=np.float32)
    bounded = bounds is not None and tuple(bounds) != (None, None)
    lo, hi = (bounds if bounded else (0.0, 0.0))
    l2 = (l2, l2) if not isinstance(l2, (tuple, list)) else l2
    mask = sum(1 << int(f) for f in set(factor))
    c_double = ctypes.c_double
    rc = lib().lrf_oracle_bcd_ex_eps(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), c_int(num_iters), c_int(int(bounded)), c_float(lo),
                                     c_float(hi), c_double(l2[0]), c_double(l2[1]), c_double(l1_ratio), c_int(mask), _ptr(U, _fp),
                                     _ptr(V, _fp), _ptr(W, _fp), c_double(eps))
    assert rc == 0
    return U, V, W


def qmf_decompose(X, R, num_iters=10, bounds=(-16, 15), sign=None):
    X = _f32(X)
    M, N = X.shape
    U = np.empty((M, R), np.float32)
    V = np.empty((N, R), np.float32)
    bounded = bounds is not None and tuple(bounds) != (None, None)
    lo, hi = (bounds if bounded else (0.0, 0.0))
    keep, sp = _sign_arg(sign, R)
    rc = lib().lrf_oracle_qmf_decompose(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), c_int(num_iters),
                                        c_int(int(bounded)), c_float(lo), c_float(hi), sp, _ptr(U, _fp), _ptr(V, _fp))
    assert rc == 0
    return U, V


def planes_to_rgb(U, V, H, W, p=8, q=8):
    """int8 factor lists (3 planes) -> uint8 [3,H,W]."""
    U = [np.ascontiguousarray(u, dtype=np.int8) for u in U]
    V = [np.ascontiguousarray(v, dtype=np.int8) for v in V]
    R = (c_int * 3)(*[u.shape[1] for u in U])
    Up = (_i8p * 3)(*[_ptr(u, _i8p) for u in U])
    Vp = (_i8p * 3)(*[_ptr(v, _i8p) for v in V])
    out = np.empty((3, H, W), np.uint8)
    rc = lib().lrf_oracle_planes_to_rgb(Up, Vp, R, c_long(H), c_long(W), c_long(p), c_long(q), _ptr(out, _u8p))
    assert rc == 0
    return out


def pad_patchify(img, p=8, q=8):
    """fp32 [C,H,W] -> X [M, C*p*q] (reflect pad + patchify)."""
    img = _f32(img)
    C, H, W = img.shape
    hp = H + (p - H % p) % p
    wp = W + (q - W % q) % q
    X = np.empty(((hp // p) * (wp // q), C * p * q), np.float32)
    lib().lrf_oracle_pad_patchify(_ptr(img, _fp), c_long(C), c_long(H), c_long(W), c_long(p), c_long(q), _ptr(X, _fp))
    return X


def depatchify_unpad(X, C, Hp, Wp, H, W, p=8, q=8):
    X = _f32(X)
    out = np.empty((C, H, W), np.float32)
    lib().lrf_oracle_depatchify_unpad(_ptr(X, _fp), c_long(C), c_long(Hp), c_long(Wp), c_long(H), c_long(W),
                                      c_long(p), c_long(q), _ptr(out, _fp))
    return out


def quantize_u8(t):
    t = _f32(t)
    q = np.empty(t.shape, np.uint8)
    sc, mn = c_float(), c_float()
    lib().lrf_oracle_quantize_u8(_ptr(t, _fp), c_long(t.size), _ptr(q, _u8p), ctypes.byref(sc), ctypes.byref(mn))
    return q, float(sc.value), float(mn.value)


def dequantize_u8(q, scale, minv):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.empty(q.shape, np.float32)
    lib().lrf_oracle_dequantize_u8(_ptr(q, _u8p), c_long(q.size), c_float(scale), c_float(minv), _ptr(t, _fp))
    return t


def to_u8(x):
    x = _f32(x)
    out = np.empty(x.shape, np.uint8)
    lib().lrf_oracle_to_u8(_ptr(x, _fp), c_long(x.size), _ptr(out, _u8p))
    return out


def svd_topr(X, R, sign=None):
    """(u [M,R], v [N,R]) = (U sqrt(s), V sqrt(s)) of the top-R singular triplets (svd_encode's factors before quantisation)."""
    X = _f32(X)
    M, N = X.shape
    u = np.empty((M, R), np.float32)
    v = np.empty((N, R), np.float32)
    keep, sp = _sign_arg(sign, R)
    lib().lrf_oracle_svd_topr.restype = c_int
    rc = lib().lrf_oracle_svd_topr(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), sp, _ptr(u, _fp), _ptr(v, _fp))
    assert rc == 0
    return u, v


def svd_topr_u8(X, R, sign=None):
    """The same for a matrix of uint8-valued floats (svd_encode's and the RGB colour space's [M, c p q] patch matrices), as
    the GPU computes it: exact Gram matrix, then the restated eigen-solver (lrf_oracle_any.c) — bit for bit the library's."""
    X = _f32(X)
    M, N = X.shape
    u = np.empty((M, R), np.float32)
    v = np.empty((N, R), np.float32)
    keep, sp = _sign_arg(sign, R)
    lib().lrf_oracle_svd_topr_u8.restype = c_int
    rc = lib().lrf_oracle_svd_topr_u8(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), sp, _ptr(u, _fp), _ptr(v, _fp))
    assert rc == 0
    return u, v


def svd_decode_rgb(qu, qv, quant_u, quant_v, H, W):
    qu = np.ascontiguousarray(qu, dtype=np.uint8)
    qv = np.ascontiguousarray(qv, dtype=np.uint8)
    M, R = qu.shape
    out = np.empty((3, H, W), np.uint8)
    rc = lib().lrf_oracle_svd_decode_rgb(_ptr(qu, _u8p), _ptr(qv, _u8p), c_long(M), c_int(R), c_float(quant_u[0]), c_float(quant_u[1]),
                                         c_float(quant_v[0]), c_float(quant_v[1]), c_long(H), c_long(W), _ptr(out, _u8p))
    assert rc == 0
    return out


# ---- qmf_encode / qmf_decode, RGB colour-space branch (lrf/compression/qmf.py:164-187, 309-323) -----------------
def qmf_rgbspace_decompose(img_u8, R, num_iters=10, bounds=(-16, 15), sign=None, init=None):
    """uint8 [3,H,W] -> int8 (u [M,R], v [192,R]): X = patchify(pad(img)), SVD initialisation (svd_topr) unless
    `init` = (u0, v0) is given, then the same BCD as the 64-column path (lrf_oracle_bcd is written for any N)."""
    X = pad_patchify(np.asarray(img_u8, dtype=np.float32))
    u0, v0 = init if init is not None else svd_topr_u8(X, R, sign)
    return bcd(X, u0, v0, num_iters, bounds)


def qmf_rgbspace_decode(u, v, H, W):
    """int8 factors -> uint8 [3,H,W]: u @ v.mT (exact integers), depatchify, unpad, clamp + truncate."""
    u = np.asarray(u, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    Hp, Wp = H + (8 - H % 8) % 8, W + (8 - W % 8) % 8
    return to_u8(depatchify_unpad(u @ v.T, 3, Hp, Wp, H, W))


# ---- qmf_encode / qmf_decode, YCbCr branch with any patch size or patch=False (qmf.py:227-286, 325-351) ----------
def svd_topr_any(X, R, sign=None, jacobi=False):
    """(u0 [M,R], v0 [N,R]) for any shape: eigen-problem on the short side.  The default is the restatement of the GPU's
    own initialisation (lrf_oracle_any.c: bit for bit the library's); jacobi=True the independent cyclic-Jacobi solver."""
    X = _f32(X)
    M, N = X.shape
    u = np.empty((M, R), np.float32)
    v = np.empty((N, R), np.float32)
    keep, sp = _sign_arg(sign, R)
    fn = lib().lrf_oracle_svd_topr_any_jacobi if jacobi else lib().lrf_oracle_svd_topr_any
    fn.restype = c_int
    rc = fn(_ptr(X, _fp), c_long(M), c_long(N), c_int(R), sp, _ptr(u, _fp), _ptr(v, _fp))
    assert rc == 0
    return u, v


def ycbcr_planes(rgb, chroma=None):
    """uint8 [3,H,W] -> [Y [H,W], Cb [h,w], Cr [h,w]] fp32: rgb_to_ycbcr + chroma_downsampling(area) to `chroma` = (h, w)
    (None: scale_factor (0.5, 0.5), i.e. floor(H / 2) x floor(W / 2); F.interpolate's area mode only sees the two sizes)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    ycc = np.empty((3, H, W), np.float32)
    lib().lrf_oracle_rgb_to_ycbcr(_ptr(rgb, _u8p), c_long(H), c_long(W), _ptr(ycc, _fp))
    h, w = (H // 2, W // 2) if chroma is None else chroma
    out = [ycc[0].copy()]
    for c in (1, 2):
        ds = np.empty((h, w), np.float32)
        lib().lrf_oracle_area_downsample(_ptr(np.ascontiguousarray(ycc[c]), _fp), c_long(H), c_long(W), c_long(h), c_long(w), _ptr(ds, _fp))
        out.append(ds)
    return out


def anyshape_matrices(rgb, patch_size, chroma=None):
    """The three matrices qmf_encode factorises: patches (p, q) of the reflect-padded planes, or the planes (None)."""
    planes = ycbcr_planes(rgb, chroma)
    if patch_size is None:
        return planes
    return [pad_patchify(pl[None], patch_size[0], patch_size[1]) for pl in planes]


def qmf_anyshape_decompose(rgb, patch_size, ranks, num_iters=10, bounds=(-16, 15), signs=None, inits=None, chroma=None):
    """-> [(u, v)] * 3 fp32 (integer valued for num_iters >= 1)."""
    out = []
    for c, X in enumerate(anyshape_matrices(rgb, patch_size, chroma)):
        u0, v0 = inits[c] if inits is not None else svd_topr_any(X, ranks[c], None if signs is None else signs[c])
        out.append(bcd(X, u0, v0, num_iters, bounds) if num_iters > 0 else (u0, v0))
    return out


def qmf_anyshape_decode(factors, H, W, patch_size, chroma=None):
    """[(u, v)] * 3 integer factors -> uint8 [3,H,W]: u @ v.mT, depatchify + unpad, nearest up-sampling, ycbcr_to_rgb, to_dtype."""
    ycc = np.empty((3, H, W), np.float32)
    for c, (u, v) in enumerate(factors):
        h, w = (H, W) if c == 0 else ((H // 2, W // 2) if chroma is None else chroma)
        X = np.asarray(u, dtype=np.float32) @ np.asarray(v, dtype=np.float32).T  # exact small integers
        if patch_size is not None:
            p, q = patch_size
            hp, wp = h + (p - h % p) % p, w + (q - w % q) % q
            pl = depatchify_unpad(X, 1, hp, wp, h, w, p, q)[0]
        else:
            pl = np.ascontiguousarray(X, dtype=np.float32)
        if c == 0:
            ycc[0] = pl
        else:
            up = np.empty((H, W), np.float32)
            lib().lrf_oracle_nearest_upsample(_ptr(np.ascontiguousarray(pl), _fp), c_long(h), c_long(w), c_long(H), c_long(W), _ptr(up, _fp))
            ycc[c] = up
    out = np.empty((3, H, W), np.float32)
    lib().lrf_oracle_ycbcr_to_rgb(_ptr(ycc, _fp), c_long(H), c_long(W), _ptr(out, _fp))
    return to_u8(out)


# ---- RGB colour-space branch for any patch size / no patches (lrf/compression/qmf.py:164-212, 309-323): pure data movement
# and exact integer arithmetic, restated in numpy
def rgb_matrix_any(rgb, patch_size):
    """uint8 [3,H,W] -> fp32 X: [M, 3 p q] for patches (p, q) (reflect pad with top = pad // 2, utils.py:108-132; patchify
    "c (h p) (w q) -> (h w) (c p q)", qmf.py:43-56), or the planes themselves [3, H, W] for patch_size None."""
    x = np.asarray(rgb, dtype=np.float32)
    if patch_size is None:
        return np.ascontiguousarray(x)
    p, q = patch_size
    _, H, W = x.shape
    ph, pw = (p - H % p) % p, (q - W % q) % q
    x = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)), mode="reflect")
    c, Hp, Wp = x.shape
    x = x.reshape(c, Hp // p, p, Wp // q, q).transpose(1, 3, 0, 2, 4)  # (h, w, c, p, q)
    return np.ascontiguousarray(x.reshape((Hp // p) * (Wp // q), c * p * q))


def rgb_decode_any(u, v, H, W, patch_size):
    """int8 factors -> uint8 [3,H,W]: u @ v.mT (exact integers), depatchify + centre crop (patches), clamp, truncate."""
    if patch_size is None:
        x = np.einsum("chr,cwr->chw", u.astype(np.int64), v.astype(np.int64))
    else:
        p, q = patch_size
        ph, pw = (p - H % p) % p, (q - W % q) % q
        Hp, Wp = H + ph, W + pw
        x = (u.astype(np.int64) @ v.astype(np.int64).T).reshape(Hp // p, Wp // q, 3, p, q).transpose(2, 0, 3, 1, 4).reshape(3, Hp, Wp)
        x = x[:, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W]
    return np.clip(x, 0, 255).astype(np.uint8)
