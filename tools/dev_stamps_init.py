"""Diagnostic: cycle shares of the phases of one Householder step of k_init's tridiagonalisation, wave 0 of each workgroup, from
a -DLRF_INIT_STAMPS build (lrf_amd/liblrf_hip_stamps.so, never the shipped library):
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DLRF_INIT_STAMPS -o lrf_amd/liblrf_hip_stamps.so lrf_amd/csrc/lrf_api.hip"""
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
    lrf_amd.qmf_factorize_batch(imgs, (7, 3, 3))
torch.cuda.synchronize()
ctx = _lib.context(0)
n = 3 * B
buf = np.zeros((n, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
tot = buf[:, 0].astype(np.float64)
print(f"matrices {n}: tridiagonalisation, s_memtime ticks (100 MHz) per matrix: median {np.median(tot):.0f}, per step {np.median(tot)/62:.1f}")
for i, name in enumerate(("column norm / reflector (owner wave) + barrier", "matvec partials + barrier", "p, K, w (wave 0) + barrier", "rank-2 update")):
    v = buf[:, 1 + i].astype(np.float64)
    print(f"  {name:48s} per step {np.median(v)/62:7.2f} ticks  share {100*np.median(v/tot):5.1f} %")
