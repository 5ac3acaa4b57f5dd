"""Host-side profile of one-image-per-call lrf_amd.qmf_encode / qmf_decode (cProfile, 200 calls each).  Development aid."""
import cProfile, os, pstats, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, lrf_amd
g = torch.Generator().manual_seed(0)
base = torch.rand(1, 3, 64, 96, generator=g) * 255
img = (torch.nn.functional.interpolate(base, size=(512, 768), mode="bilinear")[0] + torch.randn(3, 512, 768, generator=g) * 4).clamp(0, 255).to(torch.uint8)
q = float(sys.argv[1]) if len(sys.argv) > 1 else 7
for _ in range(5): enc = lrf_amd.qmf_encode(img, quality=q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): enc = lrf_amd.qmf_encode(img, quality=q)
print(f"encode {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per call, {len(enc)} bytes")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): enc = lrf_amd.qmf_encode(img, quality=q)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
for _ in range(5): lrf_amd.qmf_decode(enc)
t0 = time.perf_counter()
for _ in range(200): lrf_amd.qmf_decode(enc)
print(f"decode {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): lrf_amd.qmf_decode(enc)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
