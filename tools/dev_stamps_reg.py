"""Phase shares of k_any_tridiag_reg (the register-resident tridiagonalisation of 64 < n <= 192: svd_encode's [M,192] Gram
matrices) from s_memtime stamps of every wave's lane 0 — a library built with -DLRF_REG_STAMPS (LRF_LIB names it); development
aid.  python tools/dev_stamps_reg.py B M N R"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
import numpy as np, torch
from lrf_amd import _lib
B, M, N, R = (int(a) for a in sys.argv[1:5])
X = torch.rand(B, M, N, device="cuda") * 255
ctx = _lib.context(0)
ctx.svd_init(X, R); torch.cuda.synchronize()
ctx.svd_init(X, R); torch.cuda.synchronize()
nb = min(B, 1024)
buf = np.zeros((nb, 16, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), nb * 16 * 8) == 0
nw = 4 * ((N + 63) // 64)
tot = buf[:, :nw, 0].astype(np.float64)
print(f"B,M,N,R={(B, M, N, R)}: {np.median(tot):.0f} ticks per matrix (median over waves), {np.median(tot) / (N - 2):.0f} per step")
names = ("publish row k, barrier", "xrow read, sigma tree, sqrt, 1/x, v", "v_k store, matvec chains, partial write", "barrier, combine duty (p once per column), barrier", "p read, K tree, w", "rank-2 update")
for i, name in enumerate(names):
    v = buf[:, :nw, 1 + i].astype(np.float64)
    print(f"  {name:45s} {np.median(v) / (N - 2):7.0f} ticks per step, share {100 * np.median(v / tot):5.1f} %   (per wave medians: {' '.join('%.0f' % (np.median(v[:, w]) / (N - 2)) for w in range(nw))})")
