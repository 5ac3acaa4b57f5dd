"""Stage cycles of k_any_eig per wave (s_memtime stamps of every wave's lane 0) — a library built with -DLRF_REG_STAMPS (LRF_LIB
names it); development aid.  python tools/dev_stamps_eig.py B M N R"""
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
st = buf[:, 12:16, :7].astype(np.float64)
names = ("entry (d, e, tau to LDS)", "hull, eigenvalues", "twisted factorisation", "Gram-Schmidt", "back-transformation loop", "scaling, output")
print(f"B,M,N,R={(B, M, N, R)}: k_any_eig {np.median(st[:, :, 6] - st[:, :, 0]):.0f} cycles per matrix")
for i, name in enumerate(names):
    d = st[:, :, i + 1] - st[:, :, i]
    print(f"  {name:32s} {np.median(d):9.0f} cycles (per wave medians: {' '.join('%.0f' % np.median(d[:, w]) for w in range(4))})")
