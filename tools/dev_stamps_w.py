"""Diagnostic: per-phase cycle shares of k_bcd_w from the -DLRF_STAMPS build (never the shipped library)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "lrf_amd", "liblrf_hip_stamps.so")
import lrf_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randint(0, 256, (B, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(2):
    U, V = lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
ctx = _lib.context(0)
n = min(16384, B * 24)
buf = np.zeros((n, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
cols = [buf[:, i].astype(np.float64) for i in range(8)]
tot = cols[0]
print(f"blocks {n}: total cycles/wave median {np.median(tot):.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
names = ["prefetch wait", "LDS stores + transposes", "U mfma + shuffle", "old u + prefetch issue", "gauss-seidel", "u->LDS + int8 stores", "P/Q mfma"]
for name, v in zip(names, cols[1:]):
    print(f"  {name:26s} per sub-tile {np.median(v)/6:8.0f}   share {np.median(v / tot) * 100:5.1f}%")
