"""Single-image latency of lrf_amd.qmf_encode / qmf_decode (the way the reference's experiments call them), 512x768:
wall-clock per call and the share of the host-side container packing.  Development aid."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
from lrf_amd import codec
g = torch.Generator().manual_seed(0)
base = torch.rand(1, 3, 64, 96, generator=g) * 255
img = (torch.nn.functional.interpolate(base, size=(512, 768), mode="bilinear")[0] + torch.randn(3, 512, 768, generator=g) * 4).clamp(0, 255).to(torch.uint8)
for q in (7, 20, 32):
    for _ in range(3): enc = lrf_amd.qmf_encode(img, quality=q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): enc = lrf_amd.qmf_encode(img, quality=q)
    te = (time.perf_counter() - t0) / 10
    dev = img.cuda().unsqueeze(0)
    ranks = lrf_amd.qmf_ranks((512, 768), None, q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        U, V = lrf_amd.qmf_factorize_batch(dev, ranks); U.cpu(); V.cpu()
    tf = (time.perf_counter() - t0) / 10
    for _ in range(3): lrf_amd.qmf_decode(enc)
    t0 = time.perf_counter()
    for _ in range(10): dec = lrf_amd.qmf_decode(enc)
    td = (time.perf_counter() - t0) / 10
    print(f"quality {q} ranks {ranks}: encode {te*1e3:.2f} ms (factorisation + D2H {tf*1e3:.2f} ms, rest = H2D + container), "
          f"decode {td*1e3:.2f} ms, {len(enc)} bytes", flush=True)
