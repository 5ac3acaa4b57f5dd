"""Diagnostic: per-phase cycle shares of k_bcd from the -DLRF_STAMPS build (never the shipped library)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from lrf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "lrf_amd", "liblrf_hip_stamps.so")
import lrf_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
GSPROBE = os.environ.get("LRF_GS_PROBE") == "1"  # library built with -DLRF_GS_PROBE=1
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
tot, st, um, gs, pq = [buf[:, i].astype(np.float64) for i in range(5)]
print(f"blocks {n}: total cycles/WG median {np.median(tot):.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
for name, v in (("stage+barriers", st), ("U mfma", um), ("gauss-seidel", gs), ("P/Q mfma + U store", pq)):
    print(f"  {name:22s} median {np.median(v):8.0f}  share {np.median(v / tot) * 100:5.1f}%  per sub-tile {np.median(v)/6:.0f}")
print("  other (prologue/epilogue) share %.1f%%" % (100 * np.median((tot - st - um - gs - pq) / tot)))
g1, g2, g3 = [buf[:, i].astype(np.float64) for i in (5, 6, 7)]
print(f"  U phase split per sub-tile: wait+transpose {np.median(st)/6:.0f}  mfma chain {np.median(g1)/6:.0f}  px/commit/prefetch issue {np.median(g2)/6:.0f}  barrier {np.median(g3)/6:.0f}")
if GSPROBE:
    print(f"  (GS probe build) per GS call: LDS operand wait {np.median(st)/6:.0f}  table s_load wait {np.median(g1)/6:.0f}  speculative solve {np.median(g2)/6:.0f}  gs_row total {np.median(g3)/6:.0f}")
