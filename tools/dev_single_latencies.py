"""Single-image latencies of every encode / decode branch (host tensor in, bytes out; development aid)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lrf_amd
from conftest import config3_image
img = config3_image(3)
def t(fn, n=10):
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
for name, enc, dec in (
        ("qmf 8x8 q7", lambda: lrf_amd.qmf_encode(img, quality=7), lrf_amd.qmf_decode),
        ("qmf 16x16 q20", lambda: lrf_amd.qmf_encode(img, quality=20, patch_size=(16, 16)), lrf_amd.qmf_decode),
        ("qmf 4x4 q20", lambda: lrf_amd.qmf_encode(img, quality=20, patch_size=(4, 4)), lrf_amd.qmf_decode),
        ("qmf no patches q20", lambda: lrf_amd.qmf_encode(img, quality=20, patch=False), lrf_amd.qmf_decode),
        ("qmf RGB space q10", lambda: lrf_amd.qmf_encode(img, quality=10, color_space="RGB"), lrf_amd.qmf_decode),
        ("svd q2.5", lambda: lrf_amd.svd_encode(img, quality=2.5), lrf_amd.svd_decode)):
    te, s = t(enc)
    td, _ = t(lambda: dec(s))
    print(f"{name:20s} encode {te:7.2f} ms  decode {td:6.2f} ms  {len(s)} bytes", flush=True)
