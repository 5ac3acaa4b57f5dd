"""Diagnostic: per-phase cycle shares of k_bcd_w32 from a -DLRF_STAMPS -DLRF_W32_STAMPS build (lrf_amd/liblrf_hip_stamps32.so,
never the shipped library).  usage: python tools/dev_stamps_w32.py [ranks, default 20,10,10] [images, default 256]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "lrf_amd", "liblrf_hip_stamps32.so")
import lrf_amd
RANKS = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (20, 10, 10)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(2):
    U, V = lrf_amd.qmf_factorize_batch(imgs, RANKS)
torch.cuda.synchronize()
ctx = _lib.context(0)
n = min(16384, B * 16)  # the luma blocks come first
buf = np.zeros((n, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
cols = [buf[:, i].astype(np.float64) for i in range(8)]
tot = cols[0]
print(f"ranks {RANKS}, blocks {n}: total cycles/wave median {np.median(tot):.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
names = ["prefetch wait", "LDS stores + old row + issue", "a = x V (128 MFMA)", "tiles -> rows", "gauss-seidel + pack", "u->LDS + int8 stores", "P/Q mfma"]
acc = 0.0
for name, v in zip(names, cols[1:]):
    acc += np.median(v)
    print(f"  {name:30s} per sub-tile {np.median(v)/6:8.0f}   share {np.median(v / tot) * 100:5.1f}%")
print(f"  (start-up + epilogue per block: {np.median(tot) - acc:.0f})")
