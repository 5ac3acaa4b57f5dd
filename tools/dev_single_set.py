"""Diagnostic: single-image qmf_encode latency over the config-3 image set at one quality, split into factorisation (device
tensor in, factors back on the host) and the rest (upload + zlib container)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, lrf_amd
from conftest import config3_image
q = int(sys.argv[1]) if len(sys.argv) > 1 else 7
imgs = [config3_image(i) for i in range(24)]
ranks = lrf_amd.qmf_ranks((512, 768), None, q)
rows = []
for i, img in enumerate(imgs):
    for _ in range(3): enc = lrf_amd.qmf_encode(img, quality=q)
    t0 = time.perf_counter()
    for _ in range(10): enc = lrf_amd.qmf_encode(img, quality=q)
    te = (time.perf_counter() - t0) / 10
    dev = img.cuda().unsqueeze(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        U, V = lrf_amd.qmf_factorize_batch(dev, ranks); Uc, Vc = U.cpu(), V.cpu()
    tf = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): d = img.cuda(); torch.cuda.synchronize()
    th = (time.perf_counter() - t0) / 10
    rows.append((te, tf, th, len(enc)))
    print(f"image {i:2d}: encode {te*1e3:.3f} ms  factorise+D2H {tf*1e3:.3f}  upload alone {th*1e3:.3f}  bytes {len(enc)}", flush=True)
import statistics
print("mean encode %.3f ms, mean factorise+D2H %.3f, mean upload %.3f" % tuple(statistics.mean(r[k]) * 1e3 for k in range(3)))
