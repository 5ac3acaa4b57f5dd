"""Diagnostic: cycles of the phases of one 64-row block of k_gram64, per wave, from a -DLRF_GRAM_STAMPS build
(lrf_amd/liblrf_hip_stamps.so, never the shipped library):
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DLRF_GRAM_STAMPS -o lrf_amd/liblrf_hip_stamps.so lrf_amd/csrc/lrf_api.hip"""
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
n = 4 * min(4096, 6 * B)
buf = np.zeros((n, 8), np.uint64)
lib = _lib.load()
lib.lrf_debug_read_gram_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert lib.lrf_debug_read_gram_stamps(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), n * 8) == 0
buf = buf.reshape(-1, 4, 8).astype(np.float64)
nblk = 24.0  # 1536-row chunks (the first chunks of the table are luma)
for w in range(4):
    b = buf[:1024, w]
    print(f"wave {w}: cycles per 64-row block {np.median(b[:,0])/nblk:7.0f}: wait for loads {np.median(b[:,1])/nblk:6.0f}, digits {np.median(b[:,2])/nblk:6.0f}, "
          f"load issue + LDS write + barrier {np.median(b[:,3])/nblk:6.0f}, LDS reads + MFMA {np.median(b[:,4])/nblk:6.0f}")
