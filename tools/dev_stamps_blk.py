"""Phase times of k_any_tridiag_blk from s_memtime stamps (a library built with -DLRF_BLK_STAMPS; development aid).
python tools/dev_stamps_blk.py B M N R"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib
B, M, N, R = (int(a) for a in sys.argv[1:5])
X = torch.rand(B, M, N, device="cuda") * 255
ctx = _lib.context(0)
ctx.svd_init(X, R); torch.cuda.synchronize()
ctx.svd_init(X, R); torch.cuda.synchronize()
buf = np.zeros((B, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), B * 8) == 0
n = min(M, N)
tot = buf[:, 0].astype(np.float64)
print(f"B,M,N,R={(B, M, N, R)}: k_any_tridiag_blk {np.median(tot) / 100:.0f} us per matrix (median; 100 MHz ticks), per step {np.median(tot) / 100 / (n - 2):.2f} us")
for i, name in enumerate(("row k + corrections + sigma", "reflector, g / h trees, barrier", "product pass A0 v", "corrections, K, w", "panel update")):
    v = buf[:, 1 + i].astype(np.float64)
    print(f"  {name:36s} {np.median(v) / 100:9.1f} us  share {100 * np.median(v / tot):5.1f} %")
