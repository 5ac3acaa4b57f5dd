"""Phase shares of k_any_tridiag_blk / k_any_tridiag_sym (whichever the side selects) from s_memtime stamps of thread 0 — a
library built with -DLRF_BLK_STAMPS; development aid.  The tick unit of s_memtime is not assumed: read the SHARES.
python tools/dev_stamps_blk.py B M N R"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from lrf_amd import _lib as _l0
if os.environ.get("LRF_LIB"): _l0.LIB_PATH = os.path.join(os.path.dirname(_l0.LIB_PATH), os.environ["LRF_LIB"])
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
print(f"B,M,N,R={(B, M, N, R)}: tridiagonalisation {np.median(tot):.0f} ticks per matrix (median), {np.median(tot) / (n - 2):.0f} per step")
for i, name in enumerate(("row k + corrections + sigma (+ the previous step's K, w)", "reflector, g / h trees, barrier", "product pass A0 v", "corrections, K, w of a panel's last step", "panel update")):
    v = buf[:, 1 + i].astype(np.float64)
    print(f"  {name:58s} share {100 * np.median(v / tot):5.1f} %")
